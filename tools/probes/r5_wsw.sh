# the 128 x 256 wave-specialised weight-gradient tile (id 6): tests, grouped launch by kind x split, single launches, step A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wsw; O=gpurun_out/wsw
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "128x256 or wgrad_group" 2>&1 | tail -3
for k in 1 2; do for s in 1 2; do
  echo "== kind=$k split=$s"; VITAE_WGRAD_GROUP_WS=$k VITAE_WGRAD_GROUP_SPLIT=$s WG_ONLY_GROUP=1 timeout 300 python tools/wgrad_group_bench.py 2>&1 | grep "grouped"
done; done | tee $O/sweep.txt
echo "== auto"; WG_ONLY_GROUP=1 python tools/wgrad_group_bench.py 2>&1 | grep grouped | tee $O/auto.txt
timeout 600 python tools/bt_bench.py step forms=wgrad tiles=6,4,3,-1 2>&1 | grep -v amdgpu | grep "B32\|B8 dec pred\|B8 patch\|wgrad" | tee $O/single.txt | tail -40
