cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/wsw; mkdir -p $O
for k in 1 2; do
  VITAE_WGRAD_GROUP_WS=$k VITAE_WGRAD_GROUP_SPLIT=1 WG_ONLY_GROUP=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM --output-format csv -d $O/pmc_$k -- python tools/wgrad_group_bench.py > /dev/null 2>&1
  echo "== kind $k"; python tools/summarize_pmc.py $O/pmc_$k | grep -i "group"
  rm -rf $O/pmc_$k
done
