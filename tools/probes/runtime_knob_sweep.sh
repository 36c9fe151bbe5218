# In-step effect of HIP runtime (CLR) environment knobs on the graph-replayed step.  Every run is wrapped in `timeout`:
# GPU_MAX_HW_QUEUES=2 deadlocks the multi-stream step graph (the run never returns).
# Measured (ms/step, two runs each; defaults 5.56 / 5.57 on that box):
#   DEBUG_CLR_GRAPH_PACKET_CAPTURE=1  5.53 5.48   =0  5.50 5.49        (no effect)
#   HIP_FORCE_DEV_KERNARG=1           5.42 5.55   =0  6.01 6.03        (1 is already the default)
#   DEBUG_HIP_FORCE_GRAPH_QUEUES=1    5.60 5.60   =8  5.41 5.55        (no effect beyond noise)
#   GPU_MAX_HW_QUEUES=8               14.8 14.6   =2  hangs
run() { timeout 120 python bench.py --no-cpu-baseline --no-extra --steps 80 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" 2>/dev/null || echo fail; }
for e in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "X=2"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
