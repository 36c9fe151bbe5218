"""Large-shape sweep of the LDS-DMA GEMM (is the tile itself fast when launch cost no longer matters?).
Run once per tile selection, e.g. VITAE_GLDS_T128=1 / VITAE_GLDS_T128W8=1 python tools/gemm_big.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run_glds

if __name__ == '__main__':
    forms = sys.argv[1:] or ['fwd']
    for form in forms:
        for (M, N, K) in ((2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 4096), (3520, 3072, 768), (3520, 768, 3072), (7040, 2304, 768)):
            run_glds('big', form, M, N, K, iters=10, check=(M <= 4096))
