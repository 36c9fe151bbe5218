import sys, os
sys.path.insert(0, 'tools')
from bt_bench import one
for form, M, N, K in (('dgrad', 868, 512, 16384), ('dgrad', 1736, 512, 16384), ('fwd', 432, 768, 16384), ('fwd', 864, 768, 16384), ('wgrad', 16384, 512, 896), ('wgrad', 768, 16384, 448)):
    for t in (4, 3, -2, -1):
        one(form, M, N, K, t, iters=20, check=(t == 4))
