# DESIGN AID (not a measurement): step time on ONE GPU with every gradient bucket additionally occupying the communication
# stream for the time an 8-rank ring all-reduce would take at a given bus bandwidth (ddp._ModelledRing).
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 60 --profile-steps 0 $EXTRA 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('ddp_streams_on_own_hw_queues'))"; }
export VITAE_FORCE_DDP=1
echo "no model, picked streams: $(run)    unpicked: $(VITAE_DDP_PICK_STREAMS=0 run)"
for bw in 300 150; do
  echo "busbw $bw unpicked streams: $(VITAE_DDP_SIM_BUSBW=$bw VITAE_DDP_PICK_STREAMS=0 run)"
  for e in "X=1" "VITAE_ENC_CHUNKS=4" "VITAE_ENC_CHUNKS=6" "VITAE_ENC_CUTS=0,1,6,12" "VITAE_ENC_CUTS=0,2,7,12" "VITAE_ENC_CHUNKS=4 VITAE_ENC_CUTS=0,1,4,8,12" "VITAE_ENC_CHUNKS=6 VITAE_ENC_CUTS=0,1,3,5,7,9,12" "EXTRA=--grad-comm=fp32"; do
    a=$(env VITAE_DDP_SIM_BUSBW=$bw $e bash -c "$(declare -f run); run"); echo "busbw $bw  $e  $a"
  done
done
