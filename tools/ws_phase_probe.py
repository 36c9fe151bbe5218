"""Who waits for whom in the wave-specialised 128 x 128 k-loop (csrc/gemm_bt.hip: gemm_ws_body): s_memtime stamps of consumer wave 0
and producer wave 4 of every workgroup around the barriers B_4 .. B_9 (arrival = the wave has done its part of the k-tile,
departure = the barrier released it).  python tools/ws_phase_probe.py [M N K [form]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib
dev = 'cuda'


def run(M, N, K, form='fwd', tile=4):
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0), 'wgrad': (0, 0)}[form]
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((M, K) if akc else (K, M), device=dev, generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device=dev, generator=g).bfloat16()
    C = torch.empty(M, N, device=dev)
    lib.vitae_gemm_glds_set_bt_tile(tile)
    st = torch.cuda.current_stream().cuda_stream
    launch = lambda: lib.vitae_gemm_glds(akc, bkc, A.data_ptr(), K if akc else M, B.data_ptr(), K if bkc else N, C.data_ptr(), N, None, 0, M, N, K,
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
    t = dbg.view(-1, 2, 32).cpu().double()
    t = t[t[:, 0, 0] != 0]
    med = lambda x: float(x.median())
    c, p = t[:, 0], t[:, 1]
    print(f'ws128 {form} M={M} N={N} K={K}: {us:.1f} us/launch ({2.0 * M * N * K / us / 1e6:.0f} TF/s), {len(t)} workgroups; clocks (medians over workgroups):')
    print(f'  whole loop: consumer B_0 .. end {med(c[:, 20] - c[:, 0]):.0f} for {K // 64} k-tiles')
    for i in range(6):
        ca, cd = c[:, 2 + 3 * i], c[:, 3 + 3 * i]
        pi, pa, pd = p[:, 2 + 3 * i], p[:, 3 + 3 * i], p[:, 4 + 3 * i]
        line = f'  B_{i + 4}: consumer waits {med(cd - ca):5.0f}   producer: issue done -> arrival {med(pa - pi):5.0f}, waits {med(pd - pa):5.0f};  consumer arrives {med(ca - pa):+6.0f} after the producer'
        if i:
            line += f';  interval {med(cd - c[:, 3 + 3 * (i - 1)]):5.0f} (consumer), producer issue {med(pi - p[:, 4 + 3 * (i - 1)]):5.0f}'
        print(line)
    if float(c[:, 21].max()) > 0:      # variant library built with -DVITAE_WS_FINE_STAMPS=1
        names = ['rd K2,K3 issue', 'wait slices 0-1', 'mm K0,K1 (8 MFMA)', 'lgkmcnt(0)', 'barrier', 'rd next K0,K1 issue', 'mm K2,K3 (8 MFMA)']
        print('  k-tile 12, consumer wave 0: ' + ', '.join(f'{n} {med(c[:, 22 + i] - c[:, 21 + i]):.0f}' for i, n in enumerate(names)) + f'; sum {med(c[:, 28] - c[:, 21]):.0f}')


if __name__ == '__main__':
    a = sys.argv[1:]
    if a:
        run(int(a[0]), int(a[1]), int(a[2]), a[3] if len(a) > 3 else 'fwd')
    else:
        run(3520, 768, 3072)
        run(3520, 768, 3072, 'dgrad')
        run(6944, 512, 2048)
        run(128, 128, 3072)
