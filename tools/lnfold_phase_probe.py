"""Where the time of one LayerNorm-folding GEMM launch goes (s_memtime stamps per workgroup, csrc/gemm_glds.hip lnfold_loop), next
to the standalone LayerNorm + pipelined GEMM it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib, CONSTS

dev = 'cuda'
st = torch.cuda.current_stream().cuda_stream


def run(name, M, N, K, epi=0):
    g = torch.Generator(device=dev).manual_seed(0)
    Mp = (M + 63) // 64 * 64
    x = torch.randn(M, K, device=dev, generator=g)
    xs = x.reshape(M, K // 64, 64).permute(1, 0, 2)
    stats = torch.stack([xs.sum(2), (xs * xs).sum(2)], 2).contiguous()
    gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    w16 = (torch.randn(N, K, device=dev, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    y = torch.empty(M, N, device=dev)
    aux = torch.empty(M, N, device=dev) if epi else None
    ln16 = torch.zeros(Mp, K, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ws = torch.zeros(1 << 22, device=dev)

    def fold():
        lib.vitae_gemm_glds_lnfold(x.data_ptr(), K, stats.data_ptr(), K // 64, gamma.data_ptr(), beta.data_ptr(), 1e-6, w16.data_ptr(), K,
                                   y.data_ptr(), N, None, 0, M, N, K, bias.data_ptr(), epi, None if aux is None else aux.data_ptr(), N,
                                   ln16.data_ptr(), K, mean.data_ptr(), rstd.data_ptr(), st)

    def plain():
        lib.vitae_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, ln16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, K, 1e-6, st)
        lib.vitae_gemm_glds(1, 1, ln16.data_ptr(), K, w16.data_ptr(), K, y.data_ptr(), N, None, 0, M, N, K, bias.data_ptr(), None, 0, epi,
                            None if aux is None else aux.data_ptr(), N, 0, 1, ws.data_ptr(), None, st)

    res = {}
    for nm, f in (('fold', fold), ('ln+gemm', plain)):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            f()
        b.record(); torch.cuda.synchronize()
        res[nm] = a.elapsed_time(b) * 10
    dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    lib.vitae_gemm_glds_set_debug(dbg.data_ptr())
    fold(); torch.cuda.synchronize()
    lib.vitae_gemm_glds_set_debug(None)
    t = dbg.view(-1, 16).cpu()
    t = t[t[:, 0] != 0]
    med = lambda v: float(v.double().median())
    names = ['issue', 'first loads + gamma/beta/stats', 'X0 X1 -> LDS, barrier, frags', 'steps 0-1', 'steady loop', 'tail 4 steps', 'epilogue+drain']
    parts = ', '.join(f'{n} {med(t[:, i + 1] - t[:, i]):.0f}' for i, n in enumerate(names))
    print(f'{name} M={M} N={N} K={K}: back-to-back {res["fold"]:.1f} us vs {res["ln+gemm"]:.1f} us (LayerNorm + GEMM); {len(t)} WGs, span '
          f'{float(t[:, 7].max() - t[:, 0].min()):.0f} ticks, start skew {float(t[:, 0].max() - t[:, 0].min()):.0f}; per WG: {parts}')


run('enc qkv', 440, 2304, 768)
run('enc fc1', 440, 3072, 768, CONSTS['VITAE_EPI_GELU'])
run('dec qkv', 868, 1536, 512)
run('dec fc1', 868, 2048, 512, CONSTS['VITAE_EPI_GELU'])
