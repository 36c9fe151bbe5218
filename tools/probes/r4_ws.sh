#!/bin/bash
# wave-specialised 128x128 tile (id 4): forced-tile tests, then every family on the step's shapes at batch 8 / 32
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_bt or big_tiles" 2>&1 | tail -5 > gpurun_out/ws_tests.txt
cat gpurun_out/ws_tests.txt
timeout 900 python tools/bt_bench.py step forms=${FORMS:-fwd,dgrad,wgrad} tiles=${TILES:-4,3,0,-2} > gpurun_out/ws_bench.txt 2>&1
grep -v "^---" gpurun_out/ws_bench.txt | grep "!!!" | head
tail -n 150 gpurun_out/ws_bench.txt
