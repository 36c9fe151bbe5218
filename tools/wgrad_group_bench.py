"""One grouped launch of a block's four weight gradients (vitae_wgrad_group_bt) against four separate launches through
vitae_linear_bwd_pair_glds' planner (here: vitae_gemm_glds in the dy^T x form), batch-32 / patch-8 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vit_ae_plus_plus_amd._abi import lib
from bt_bench import graph_time
dev = 'cuda'
ws = torch.zeros(1 << 24, device=dev)
for name, M, dims in (('B32 enc', 3520, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
                      ('B32 dec', 6944, [(1536, 512), (512, 512), (2048, 512), (512, 2048)]),
                      ('p8 enc', 3464, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
                      ('p8 dec', 6916, [(1536, 512), (512, 512), (2048, 512), (512, 2048)]),
                      ('B16 enc', 1760, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
                      ('B16 dec', 3472, [(1536, 512), (512, 512), (2048, 512), (512, 2048)]),
                      ('B8 enc', 880, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
                      ('B8 dec', 1736, [(1536, 512), (512, 512), (2048, 512), (512, 2048)])):
    Mp = (M + 63) // 64 * 64
    dys = [torch.randn(Mp, N, device=dev).bfloat16() for N, K in dims]
    xs = [torch.randn(Mp, K, device=dev).bfloat16() for N, K in dims]
    dws = [torch.empty(N, K, device=dev) for N, K in dims]
    arr = lambda ts: np.array([t.data_ptr() for t in ts], dtype=np.uint64)
    a_dy, a_x, a_dw = arr(dys), arr(xs), arr(dws)
    Ns, Ks = np.array([d[0] for d in dims], dtype=np.int32), np.array([d[1] for d in dims], dtype=np.int32)
    stf = lambda: torch.cuda.current_stream().cuda_stream
    grp = lambda: lib.vitae_wgrad_group_bt(4, a_dy.ctypes.data, a_x.ctypes.data, a_dw.ctypes.data, None, None, Ns.ctypes.data, Ks.ctypes.data, M, Mp, 0,
                                           ws.data_ptr(), ws.numel(), stf())
    def sep():
        for i, (N, K) in enumerate(dims):
            sp = lib.vitae_gemm_glds_pick_split_k(N, K, Mp)
            lib.vitae_gemm_glds(0, 0, dys[i].data_ptr(), N, xs[i].data_ptr(), K, dws[i].data_ptr(), K, None, 0, N, K, Mp, None, None, 0, 0, None, 0, 0, sp,
                                ws.data_ptr(), None, stf())
    if os.environ.get('WG_ONLY_GROUP'):          # kind / split sweeps: VITAE_WGRAD_GROUP_WS, VITAE_WGRAD_GROUP_SPLIT in the environment
        if grp() != 0:
            print(f'{name}: not served'); continue
        torch.cuda.synchronize()
        print(f'{name}: grouped {graph_time(grp, 10):6.1f} us', flush=True)
        continue
    grp(); sep(); torch.cuda.synchronize()
    print(f'{name}: grouped {graph_time(grp, 10):6.1f} us   four launches {graph_time(sep, 10):6.1f} us', flush=True)
