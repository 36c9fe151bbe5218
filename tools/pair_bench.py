"""The backward launches of a transformer block's Linears as the batch-4 / batch-8 step issues them (round 6):
vitae_linear_bwd_pair_glds (input gradient + weight gradient in ONE launch) with the step's real epilogues — GELU' from the saved
bf16 derivative on fc2, the bias gradients as row sums in the weight-gradient workgroups, the gradient norm's share — next to the
same halves as launches of their own.
    python tools/pair_bench.py [B=4] [iters=20] [model=L128] [tile=5]
Prints us per launch (graph replay of `iters` back-to-back launches, 4 operand sets cycled) and the launch's algorithmic TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib, CONSTS
from tools.bt_bench import graph_time

dev = 'cuda'
args = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
B = int(args.get('B', 4))
iters = int(args.get('iters', 20))
SQ = int(args.get('sq', 2))      # the gradient norm's share of the weight-gradient workgroups: 0 none, 1 ONE address (round 5), 2 spread slots
DGELU, A16, AD = CONSTS['VITAE_EPI_DGELU'], CONSTS['VITAE_EPI_AUX_BF16'], CONSTS['VITAE_EPI_AUX_DERIV']
P = lambda t: None if t is None else t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream


def case(name, M, N, K, gelu=False, dx32=True, dx16=True, dycs=False):
    """y = x W^T: x [M, K], W [N, K], dy [M, N]  ->  dx [M, K] = epi(dy W), dW [N, K] = dy^T x"""
    Mp = (M + 63) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    mk = lambda r, c: [torch.randn(r, c, device=dev, generator=g).bfloat16() for _ in range(4)]
    dys, ws_, xs = mk(Mp, N), mk(N, K), mk(Mp, K)
    for t in dys + xs:
        t[M:].zero_()
    aux = torch.rand(M, K, device=dev, generator=g).bfloat16() if gelu else None
    dx = torch.empty(M, K, device=dev) if dx32 else None
    dx16_ = torch.empty(Mp, K, device=dev, dtype=torch.bfloat16) if dx16 else None
    dw = torch.empty(N, K, device=dev)
    cs = torch.zeros(N, device=dev) if dycs else None
    acc = torch.zeros(CONSTS['VITAE_ACC_COUNT'], device=dev, dtype=torch.float64)
    ws = torch.zeros(1 << 24, device=dev)
    split = lib.vitae_linear_bwd_pair_pick_split_k(M, Mp, N, K)
    epi = (DGELU | A16 | AD) if gelu else 0
    cnt = [0]

    def pair():
        cnt[0] += 1; i = cnt[0] % 4
        lib.vitae_linear_bwd_pair_glds(P(dys[i]), P(ws_[i]), P(xs[i]), P(dx), P(dx16_), P(dw), None, M, Mp, N, K, epi, P(aux), None, P(cs),
                                       0, 0, split, P(ws), ws.numel(), st())

    def dgrad():
        cnt[0] += 1; i = cnt[0] % 4
        lib.vitae_linear_bwd_pair_glds(P(dys[i]), P(ws_[i]), None, P(dx), P(dx16_), None, None, M, Mp, N, K, epi, P(aux), None, None,
                                       0, 0, split, P(ws), ws.numel(), st())

    def wgrad():
        cnt[0] += 1; i = cnt[0] % 4
        lib.vitae_gemm_glds(0, 0, P(dys[i]), N, P(xs[i]), K, P(dw), K, None, K, N, K, Mp, None, None, 0, 0, None, 0, 0, 1, None, None, st())

    if SQ == 2:
        lib.vitae_gemm_glds_set_wgrad_sqnorm_spread(acc.data_ptr() + 8 * CONSTS['VITAE_ACC_SQ_BASE'], CONSTS['VITAE_ACC_SQ_SLOTS'], CONSTS['VITAE_ACC_SQ_STRIDE'])
    else:
        lib.vitae_gemm_glds_set_wgrad_sqnorm(acc.data_ptr() + 8 * CONSTS['VITAE_ACC_GRADSQ'] if SQ else None)
    out = []
    for f in (pair, dgrad, wgrad):
        f(); torch.cuda.synchronize()
        out.append(graph_time(f, iters))
    lib.vitae_gemm_glds_set_wgrad_sqnorm(None)
    fl = 4e-6 * M * N * K
    print(f'{name:14s} M={M:5d} N={N:5d} K={K:5d} split {split} | pair {out[0]:6.1f} us ({fl / out[0]:5.0f} TF/s) | dgrad alone {out[1]:6.1f} | '
          f'wgrad alone {out[2]:6.1f} | sum {out[1] + out[2]:6.1f}', flush=True)
    return out[0]


tot = 0.0
Me, Md = 2 * B * 55, B * 217
shapes = (('enc', Me, 768, 3072, 12), ('dec', Md, 512, 2048, 8))
if args.get('model') == 'L128':       # BASELINE config 4: ViT-L/16 on 128^3 volumes (512 patches, 25 % kept + cls; decoder 512 wide on 513 tokens)
    shapes = (('enc', B * 129, 1024, 4096, 24), ('dec', B * 513, 512, 2048, 8))
if 'tile' in args:                    # force one tile family of the planner (5 = wave-specialised 64 x 64 pair)
    lib.vitae_gemm_glds_set_bt_tile(int(args['tile']))
for pre, M, d, h, depth in shapes:
    t = 0.0
    t += case(f'{pre} fc2', M, d, h, gelu=True, dx32=False)
    t += case(f'{pre} fc1', M, h, d, dx16=False, dycs=True)
    t += case(f'{pre} proj', M, d, d, dx16=False)
    t += case(f'{pre} qkv', M, 3 * d, d, dx16=False, dycs=True)
    tot += t * depth
print(f'B={B}: pair launches of one step (12 encoder + 8 decoder blocks): {tot:.0f} us')
