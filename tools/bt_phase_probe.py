"""Where a big-tile GEMM launch (csrc/gemm_bt.hip) spends its clocks: s_memtime stamps of wave 0 (stagger group 0) and wave 4
(group 1) of every workgroup.  python tools/bt_phase_probe.py [tile] [M N K] [form]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib

dev = 'cuda'


def run(tile, M, N, K, form='fwd'):
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0), 'wgrad': (0, 0)}[form]
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((M, K) if akc else (K, M), device=dev, generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device=dev, generator=g).bfloat16()
    C = torch.empty(M, N, device=dev)
    lib.vitae_gemm_glds_set_bt_tile(tile)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        lib.vitae_gemm_glds(akc, bkc, A.data_ptr(), K if akc else M, B.data_ptr(), K if bkc else N, C.data_ptr(), N, None, 0, M, N, K,
                            None, None, 0, 0, None, 0, 0, 1, None, None, st)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        launch()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    dbg = torch.zeros(8192 * 2 * 32, dtype=torch.int64, device=dev)
    lib.vitae_gemm_glds_set_debug(dbg.data_ptr())
    launch(); torch.cuda.synchronize()
    lib.vitae_gemm_glds_set_debug(None)
    lib.vitae_gemm_glds_set_bt_tile(-1)
    t = dbg.view(-1, 2, 32).cpu()
    t = t[t[:, 0, 0] != 0]
    med = lambda x: float(x.double().median())
    print(f'tile {tile} {form} M={M} N={N} K={K}: {us:.1f} us/launch ({2.0 * M * N * K / us / 1e6:.0f} TF/s), {len(t)} workgroups, '
          f'kernel span {float(t[:, :, 25].max() - t[:, 0, 0].min()):.0f} clk, start skew {float(t[:, 0, 0].max() - t[:, 0, 0].min()):.0f}')
    for grp in range(2 if tile != 3 else 1):
        x = t[:, grp]
        d = lambda i, j: med(x[:, i] - x[:, j])
        print(f'  group {grp}: prologue issue {d(1, 0):.0f}, first wait {d(2, 1):.0f}, k-loop {d(16, 2):.0f} ({d(16, 2) / (K / 64):.0f} per k-tile), '
              f'k-tile 2: {d(14, 15):.0f} = ' +
              ' | '.join(f'P{ph + 1}: L {d(3 + 3 * ph, 15 if ph == 0 else 2 + 3 * ph):.0f} M {d(4 + 3 * ph, 3 + 3 * ph):.0f} bar {d(5 + 3 * ph, 4 + 3 * ph):.0f}' for ph in range(4)))
        print(f'           epilogue: barrier {d(17, 16):.0f} ' + ' '.join(f'q{q}: {d(18 + q, 17 + q):.0f}' for q in range(4)) +
              f' drain {d(25, 21):.0f}; total {d(25, 0):.0f}')


if __name__ == '__main__':
    a = sys.argv[1:]
    if a:
        run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), a[4] if len(a) > 4 else 'fwd')
    else:
        run(0, 440, 768, 768)
        run(0, 440, 768, 3072)
        run(0, 4096, 4096, 4096)
        run(0, 3520, 3072, 768)
        run(0, 6944, 16384, 512)
        run(3, 4096, 4096, 4096)
        run(3, 3520, 3072, 768)
        run(0, 4096, 4096, 4096, 'wgrad')
