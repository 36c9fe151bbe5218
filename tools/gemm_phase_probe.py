"""Where the time of one small GEMM launch goes: s_memtime stamps per workgroup of gemm_glds_kernel (start, prologue
issued, first tile landed, k-loop done, split-K ticket, epilogue start, epilogue stores issued, stores drained) + the
HIP-event time of the launch alone on the chip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib, CONSTS

def run(name, M, N, K, akc=1, bkc=1, res=True, epi=0, split=None, cold=False):
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    bf = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    Mp = (M + 63) // 64 * 64
    A = bf(Mp, K) if akc else bf(K, Mp)
    B = bf(N, K) if bkc else bf(K, N)
    C = torch.empty(M, N, device=dev)
    C16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    R = torch.randn(M, N, device=dev, generator=g) if res else None
    bias = torch.randn(N, device=dev, generator=g)
    aux = torch.empty(M, N, device=dev) if epi else None
    ws = torch.zeros(1 << 23, device=dev)
    sp = split or lib.vitae_gemm_glds_pick_split_k(M, N, K)
    if epi == CONSTS['VITAE_EPI_GELU']:
        sp = 1
    nwg = 4096 * max(sp, 1)
    dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def launch():
        lib.vitae_gemm_glds(akc, bkc, A.data_ptr(), K if akc else Mp, B.data_ptr(), K if bkc else N, C.data_ptr(), N, C16.data_ptr(), N, M, N, K,
                            bias.data_ptr(), None if R is None else R.data_ptr(), N, epi, None if aux is None else aux.data_ptr(), N, 0, sp,
                            ws.data_ptr(), None, st)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    trash = torch.empty(1 << 28, device=dev) if cold else None
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot, n = 0.0, 20
    for _ in range(n):
        if cold:
            trash.fill_(1.0)
        a.record(); launch(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    lib.vitae_gemm_glds_set_debug(dbg.data_ptr())
    if cold:
        trash.fill_(1.0)
    launch(); torch.cuda.synchronize()
    lib.vitae_gemm_glds_set_debug(None)
    t = dbg.view(-1, 16).cpu()
    t = t[t[:, 0] != 0]
    fin = t[t[:, 7] != 0]
    med = lambda x: float(x.double().median())
    first, last = t[:, 0].min(), (fin[:, 7].max() if len(fin) else t[:, 3].max())
    print(f'{name}: M={M} N={N} K={K} split={sp} cold={cold}: events {tot / n * 1e3:.1f} us; {len(t)} WGs ({len(fin)} finishers); '
          f'kernel span {float(last - first):.0f} clk; start skew {float(t[:, 0].max() - first):.0f}; '
          f'prologue issue {med(t[:, 1] - t[:, 0]):.0f}, first tile wait {med(t[:, 2] - t[:, 1]):.0f}, k-loop {med(t[:, 3] - t[:, 2]):.0f}'
          + f'; step 4: wait {med(t[:, 8] - t[:, 12]):.0f} barrier {med(t[:, 9] - t[:, 8]):.0f} issue {med(t[:, 10] - t[:, 9]):.0f} ds_read {med(t[:, 11] - t[:, 10]):.0f} mfma issue {med(t[:, 13] - t[:, 11]):.0f}'
          + (f', to epilogue {med(fin[:, 5] - fin[:, 3]):.0f}, epilogue {med(fin[:, 6] - fin[:, 5]):.0f} (first barrier {med(fin[:, 12] - fin[:, 5]):.0f}, '
             f'tile -> LDS + barrier {med(fin[:, 13] - fin[:, 12]):.0f}, rows out {med(fin[:, 6] - fin[:, 13]):.0f}), drain {med(fin[:, 7] - fin[:, 6]):.0f}' if len(fin) else ''))

GELU = CONSTS['VITAE_EPI_GELU']
for cold in (False,):
    run('proj fwd', 440, 768, 768, cold=cold)
    run('qkv fwd', 440, 2304, 768, res=False, cold=cold)
    run('fc1 fwd', 440, 3072, 768, res=False, epi=GELU, cold=cold)
    run('dgrad-like (KC,row)', 440, 768, 2304, akc=1, bkc=0, res=False, cold=cold)
    run('big qkv fwd B=32', 3520, 2304, 768, res=False, cold=cold)
