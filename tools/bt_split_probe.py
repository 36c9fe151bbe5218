"""One shape, forced big tile and forced split (VITAE_BT_TILE / VITAE_BT_SPLIT in the environment decide): GPU time per launch.
    VITAE_BT_TILE=3 VITAE_BT_SPLIT=1 python tools/bt_split_probe.py fwd 3520 768 3072"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bt_bench
form, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bt_bench.one(form, M, N, K, int(os.environ.get('VITAE_BT_TILE', '-1')), iters=40)
