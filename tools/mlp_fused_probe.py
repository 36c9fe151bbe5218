"""Phase timing of the fused MLP kernels (s_memtime stamps per workgroup) + wall time alone on the chip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib

def run(M, d, H, bwd, cold=False):
    Mp = (M + 63) // 64 * 64
    S = lib.vitae_mlp_fused_slabs(H)
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    bf = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    x, w1, w2 = bf(Mp, d), bf(H, d), bf(d, H)
    b1 = torch.randn(H, device=dev, generator=g)
    hpre, out16 = bf(Mp, H), bf(Mp, H)
    slabs = torch.empty(S, Mp, d, device=dev)
    nwg = 8 * ((S + 7) // 8) * (Mp // 64)
    dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def launch():
        if bwd:
            lib.vitae_mlp_fused_bwd(x.data_ptr(), w1.data_ptr(), w2.data_ptr(), hpre.data_ptr(), out16.data_ptr(), slabs.data_ptr(), M, Mp, d, H, st)
        else:
            lib.vitae_mlp_fused_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), hpre.data_ptr(), out16.data_ptr(), slabs.data_ptr(), M, Mp, d, H, st)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    trash = torch.empty(1 << 28, device=dev) if cold else None
    n = 20
    tot = 0.0
    for _ in range(n):
        if cold:
            trash.fill_(1.0)
        a.record(); launch(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    lib.vitae_mlp_fused_set_debug(dbg.data_ptr())
    if cold:
        trash.fill_(1.0)
    launch()
    torch.cuda.synchronize()
    lib.vitae_mlp_fused_set_debug(None)
    t = dbg.view(nwg, 16).cpu()
    t = t[t[:, 0] != 0]
    rel = (t[:, 1:7] - t[:, :6]).double()
    names = ['phase1', 'epi1', 'imgstore', 'p2 first', 'p2 rest', 'drain']
    span = float((t[:, 6].max() - t[:, 0].min()))
    print(f'M={M} d={d} H={H} bwd={bwd} cold={cold}: events {tot / n * 1e3:.1f} us/launch; {len(t)} WGs; span {span:.0f} ticks; '
          + ', '.join(f'{nm} {rel[:, i].median():.0f} (max {rel[:, i].max():.0f})' for i, nm in enumerate(names))
          + f'; step4: wait {float((t[:,8]-t[:,7]).double().median()):.0f} barrier {float((t[:,9]-t[:,8]).double().median()):.0f} dsread {float((t[:,10]-t[:,9]).double().median()):.0f} mfma+dma+st {float((t[:,11]-t[:,10]).double().median()):.0f}'
          + f'; start skew {float(t[:, 0].max() - t[:, 0].min()):.0f}')

if len(sys.argv) > 1:
    for spec in sys.argv[1:]:
        M, d, H, bwd = (int(v) for v in spec.split(','))
        run(M, d, H, bool(bwd))
else:
    for bwd in (False, True):
        run(440, 768, 3072, bwd)
        run(440, 768, 3072, bwd, cold=True)
        run(868, 512, 2048, bwd)
