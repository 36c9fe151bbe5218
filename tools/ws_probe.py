"""Wave-specialised 128x128 tile (id 4) against the other families on the shapes it is meant for (few output columns, long
reduction) and on two it is not; VITAE_HIP_LIB=<variant .so> for compile-time A/B.
    python tools/ws_probe.py [tiles=4,3,0,-2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bt_bench import one

SHAPES = [('fwd', 3520, 768, 3072), ('fwd', 3520, 768, 768), ('dgrad', 3520, 768, 2304), ('fwd', 6944, 512, 2048),
          ('dgrad', 6944, 512, 1536), ('fwd', 3520, 3072, 768), ('fwd', 6944, 2048, 512), ('wgrad', 3072, 768, 3520),
          ('fwd', 880, 2304, 768), ('fwd', 880, 3072, 768), ('fwd', 880, 768, 3072), ('fwd', 1736, 2048, 512), ('fwd', 1736, 512, 2048)]
if __name__ == '__main__':
    tiles = next(([int(x) for x in a.split('=')[1].split(',')] for a in sys.argv[1:] if a.startswith('tiles=')), [4])
    for form, M, N, K in SHAPES:
        for t in tiles:
            one(form, M, N, K, t, iters=20, check=(t == 4))
