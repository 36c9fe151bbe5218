cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wsw; O=gpurun_out/wsw
echo "== auto"; WG_ONLY_GROUP=1 python tools/wgrad_group_bench.py 2>&1 | grep grouped | tee $O/auto2.txt
timeout 600 python tools/bt_bench.py step forms=wgrad tiles=6,4,3,0,-1 2>&1 | grep -v amdgpu > $O/single_raw.txt; python tools/forms_table.py $O/single_raw.txt | tee $O/single.txt | grep "B32\|B8 dec pred\|B8 patch\|shape"
for r in 1 2; do for k in 1 -1; do for cfg in 32:16 4:8 8:16; do B=${cfg%%:*}; P=${cfg#*:}
  VITAE_WGRAD_GROUP_WS=$k VITAE_BT_WSW=$([ $k = 1 ] && echo 0 || echo 1) python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 30 --warmup 8 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group_kind=$k B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/ab.txt
