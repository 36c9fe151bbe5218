"""Where a wave-specialised 64 x 64 launch spends its time (round 6): s_memtime stamps of consumer wave 0 and producer wave 4 of
EVERY workgroup of one launch (csrc/gemm_bt.hip: gemm_ws64_body's `stamp`), for the paired backward launches of the batch-4 step
and their halves alone.
    python tools/ws64_phase_probe.py [B=4]
Per launch: when workgroups start (dispatch ramp), how long each phase takes (median over workgroups, shader clocks), when the
last workgroup ends; HIP-event time of the launch alone next to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib, CONSTS

dev = 'cuda'
args = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
B = int(args.get('B', 4))
DGELU, A16, AD = CONSTS['VITAE_EPI_DGELU'], CONSTS['VITAE_EPI_AUX_BF16'], CONSTS['VITAE_EPI_AUX_DERIV']
P = lambda t: None if t is None else t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream


def q(x, f):
    return float(torch.quantile(x.double(), f)) if len(x) else float('nan')


def report(name, launch, nwg_max=8192):
    dbg = torch.zeros(nwg_max * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(10):
        a.record(); launch(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    lib.vitae_gemm_glds_set_debug(dbg.data_ptr())
    launch(); torch.cuda.synchronize()
    lib.vitae_gemm_glds_set_debug(None)
    raw = dbg.view(-1, 16).cpu()
    t = raw[raw[:, 0] != 0]
    t0 = int(t[:, [0, 8]].min())
    start = (t[:, 8] - t0)                       # producer wave's first instruction
    fin = t[t[:, 6] != 0]                        # workgroups that ran the epilogue (not the non-last split-K arrivers)
    end = torch.where(t[:, 6] != 0, t[:, 6], t[:, 3]) - t0
    clk = 2.4e3                                  # clocks per us (the kernels run at 2.4 GHz in the step; s_memtime counts shader clocks)
    print(f'{name}: events {tot / 10 * 1e3:5.1f} us | {len(t)} workgroups, span {float(end.max()) / clk:5.1f} us | starts: median {q(start, .5) / clk:4.1f} '
          f'p90 {q(start, .9) / clk:4.1f} max {float(start.max()) / clk:4.1f} us | ends: median {q(end, .5) / clk:4.1f} p90 {q(end, .9) / clk:4.1f} us')
    print(f'    producer: prologue issue {q(t[:, 9] - t[:, 8], .5):5.0f} clk, first k-tile lands {q(t[:, 10] - t[:, 9], .5):5.0f}, '
          f'k-loop {q(t[:, 11] - t[:, 10], .5):5.0f} | consumer: to B_0 {q(t[:, 1] - t[:, 0], .5):5.0f} (arrives {q(t[:, 0] - t[:, 8], .5):4.0f} after start), '
          f'k-loop {q(t[:, 2] - t[:, 1], .5):5.0f}, ' + (f'split park + ticket {q(t[:, 3] - t[:, 2], .5):5.0f}, ' if int((t[:, 3] != 0).sum()) else '') +
          f'epilogue {q(fin[:, 5] - fin[:, 4], .5):5.0f}, norm share + drain {q(fin[:, 6] - fin[:, 5], .5):5.0f}; whole workgroup {q(end - start, .5):5.0f} (p90 {q(end - start, .9):5.0f})')
    return raw, t0


def case(name, M, N, K, gelu=False, dx32=True, dx16=True, dycs=False):
    Mp = (M + 63) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    mk = lambda r, c: torch.randn(r, c, device=dev, generator=g).bfloat16()
    dy, w, x = mk(Mp, N), mk(N, K), mk(Mp, K)
    dy[M:].zero_(); x[M:].zero_()
    aux = torch.rand(M, K, device=dev, generator=g).bfloat16() if gelu else None
    dx = torch.empty(M, K, device=dev) if dx32 else None
    dx16_ = torch.empty(Mp, K, device=dev, dtype=torch.bfloat16) if dx16 else None
    dw = torch.empty(N, K, device=dev)
    cs = torch.zeros(N, device=dev) if dycs else None
    acc = torch.zeros(CONSTS['VITAE_ACC_COUNT'], device=dev, dtype=torch.float64)
    ws = torch.zeros(1 << 24, device=dev)
    split = lib.vitae_linear_bwd_pair_pick_split_k(M, Mp, N, K)
    epi = (DGELU | A16 | AD) if gelu else 0
    lib.vitae_gemm_glds_set_wgrad_sqnorm_spread(acc.data_ptr() + 8 * CONSTS['VITAE_ACC_SQ_BASE'], CONSTS['VITAE_ACC_SQ_SLOTS'], CONSTS['VITAE_ACC_SQ_STRIDE'])
    pair = lambda: lib.vitae_linear_bwd_pair_glds(P(dy), P(w), P(x), P(dx), P(dx16_), P(dw), None, M, Mp, N, K, epi, P(aux), None, P(cs), 0, 0, split,
                                                  P(ws), ws.numel(), st())
    dgrad = lambda: lib.vitae_linear_bwd_pair_glds(P(dy), P(w), None, P(dx), P(dx16_), None, None, M, Mp, N, K, epi, P(aux), None, None, 0, 0, split,
                                                   P(ws), ws.numel(), st())
    wgrad = lambda: lib.vitae_gemm_glds(0, 0, P(dy), N, P(x), K, P(dw), K, None, K, N, K, Mp, None, None, 0, 0, None, 0, 0, 1, None, None, st())
    print(f'--- {name}: M={M} N={N} K={K} (dgrad split {split})')
    t, t0 = report('  pair ', pair)
    # the two halves of the pair separately: dgrad workgroups come first in the launch
    td = (M + 63) // 64 * ((K + 63) // 64)
    nb1 = (td + 7) // 8 * 8 * split
    clk = 2.4e3
    for nm, part in (('dgrad half', t[:nb1]), ('wgrad half', t[nb1:])):
        part = part[part[:, 0] != 0]
        if len(part):
            s_ = part[:, 8] - t0
            e_ = torch.where(part[:, 6] != 0, part[:, 6], part[:, 3]) - t0
            print(f'      {nm}: {len(part)} workgroups, start median {q(s_, .5) / clk:4.1f} p90 {q(s_, .9) / clk:4.1f} max {float(s_.max()) / clk:4.1f} us, '
                  f'end median {q(e_, .5) / clk:4.1f} max {float(e_.max()) / clk:4.1f} us, k-loop {q(part[:, 2] - part[:, 1], .5):5.0f} clk, life {q(e_ - s_, .5):5.0f} clk')
    report('  dgrad', dgrad)
    report('  wgrad', wgrad)
    lib.vitae_gemm_glds_set_wgrad_sqnorm(None)


Me, Md = 2 * B * 55, B * 217
for pre, M, d, h in (('enc', Me, 768, 3072), ('dec', Md, 512, 2048)):
    case(f'{pre} fc2', M, d, h, gelu=True, dx32=False)
    case(f'{pre} fc1', M, h, d, dx16=False, dycs=True)
    case(f'{pre} proj', M, d, d, dx16=False)
    case(f'{pre} qkv', M, 3 * d, d, dx16=False, dycs=True)
