"""ws64 (tile id 5) against the 64-row family on the batch-4 / batch-8 shapes (forward, input-gradient and weight-gradient forms)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bt_bench import one, NAMES
NAMES[5] = 'ws64'
SH = [('fwd', 440, 2304, 768), ('fwd', 440, 768, 768), ('fwd', 440, 3072, 768), ('fwd', 868, 2048, 512), ('fwd', 868, 512, 512),
      ('dgrad', 440, 768, 768), ('dgrad', 868, 512, 1536),
      ('wgrad', 768, 768, 448), ('wgrad', 2304, 768, 448), ('wgrad', 3072, 768, 448), ('wgrad', 768, 3072, 448),
      ('wgrad', 512, 512, 896), ('wgrad', 1536, 512, 896), ('wgrad', 2048, 512, 896), ('wgrad', 512, 2048, 896),
      ('fwd', 880, 3072, 768), ('wgrad', 3072, 768, 896)]
for form, M, N, K in SH:
    for t in (5, -2):
        one(form, M, N, K, t, iters=20, check=(t == 5))
