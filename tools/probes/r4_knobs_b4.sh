# batch-4 step: alternating sweeps of the GEMM-selection knobs (each KNOB: its default against candidates, two rounds)
for kv in "VITAE_GLDS_PIPE_MAX_WGS:512 256 1024" "VITAE_PAIR_SPLIT_TARGET:10 6 16" "VITAE_GLDS_SPLIT_BLOCKS:384 256 512" "VITAE_GLDS_WIDE_MIN_TILES:400 300 600" "VITAE_LOSS_WGS:256 512" "VITAE_HPRE_BF16:auto 0"; do
  k=${kv%%:*}; vals=${kv#*:}
  KNOB=$k VALS="$vals" ROUNDS=2 CFGS="4:16" STEPS=60 bash tools/probes/ab.sh | grep "min "
done
