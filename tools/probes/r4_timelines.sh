# Round 4: one replayed step as a timeline, batch 32 (patch 16) and patch 8 (batch 4) — where the critical path of the
# compute-bound points lies (tools/timeline.py), plus their per-kernel tables.  Step 30 of the trace: past graph priming.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4tl; mkdir -p $O
for cfg in ${CFGS:-32:16 4:8}; do
  B=${cfg%%:*}; P=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 30 --warmup 5 --profile-steps 0 > $O/stats_b${B}_p$P.log 2>&1
  python tools/timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) 30 > $O/timeline_b${B}_p$P.txt 2>&1
  python tools/prof_summary.py $O/stats 20 > $O/kernels_b${B}_p$P.txt 2>&1
  rm -rf $O/stats
  grep '^{' $O/stats_b${B}_p$P.log | tail -1 | cut -c1-200
done
