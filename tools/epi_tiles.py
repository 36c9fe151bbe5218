"""The Linears of a transformer block with the epilogues the step really runs (bf16-only outputs, GELU with the saved bf16
pre-activation, GELU' with its bf16 read and the bias-gradient column sums, residual adds), every tile family forced in turn:
    python tools/epi_tiles.py [B=32] [tiles=0,3,4,5,-2,-1]
Prints us per launch (graph replay of 20 back-to-back launches, 4 operand sets cycled) and which tile the planner picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib, CONSTS
from tools.bt_bench import graph_time, NAMES
dev = 'cuda'
args = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
B = int(args.get('B', 32))
tiles = [int(t) for t in args.get('tiles', '0,3,4,5,-2,-1').split(',')]
GELU, DGELU, A16 = CONSTS['VITAE_EPI_GELU'], CONSTS['VITAE_EPI_DGELU'], CONSTS['VITAE_EPI_AUX_BF16']
Me, Md = 2 * B * 55, B * 217
P = lambda t: None if t is None else t.data_ptr()


def case(name, form, M, N, K, out32=False, out16=False, bias=False, res=False, epi=0, colsum=False):
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0)}[form]
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    As = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(4)]
    Bs = [(torch.randn(N, K, device=dev, generator=g) if bkc else torch.randn(K, N, device=dev, generator=g)).bfloat16() for _ in range(4)]
    C = torch.empty(M, N, device=dev) if out32 else None
    C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if out16 else None
    bv = torch.randn(N, device=dev) if bias else None
    rv = torch.randn(M, N, device=dev) if res else None
    aux = torch.randn(M, N, device=dev).bfloat16() if epi else None
    cs = torch.zeros(N, device=dev) if colsum else None
    ws = torch.zeros(1 << 24, device=dev)
    row = []
    for t in tiles:
        lib.vitae_gemm_glds_set_bt_tile(t)
        got = lib.vitae_gemm_glds_bt_choice(akc, bkc, M, N, K)
        split = 1 if (epi & 15) == GELU else lib.vitae_gemm_glds_pick_split_k_form(akc, bkc, M, N, K)
        cnt = [0]

        def go():
            cnt[0] += 1; i = cnt[0] % 4
            lib.vitae_gemm_glds(akc, bkc, P(As[i]), K, P(Bs[i]), K if bkc else N, P(C), N, P(C16), N, M, N, K, P(bv), P(rv), N, epi, P(aux), N, 0,
                                split, P(ws), P(cs), torch.cuda.current_stream().cuda_stream)
        try:
            go(); torch.cuda.synchronize()
            us = graph_time(go, 20)
        except Exception as e:
            us = float('nan')
        row.append((t, got, split, us))
    lib.vitae_gemm_glds_set_bt_tile(-1)
    best = min(u for (t, g_, s, u) in row if t != -1 and u == u)
    auto = next((u for (t, g_, s, u) in row if t == -1), float('nan'))
    autot = next(((g_, s) for (t, g_, s, u) in row if t == -1), None)
    print(f'{name:22s} {form:5s} {M:5d} {N:5d} {K:5d} | ' + ' '.join(f'{NAMES[t]:>7s} {u:6.1f}{"*" if u == best else " "}' for (t, g_, s, u) in row if t != -1) +
          f' | auto {auto:6.1f} {auto / best:5.2f} {autot}  {2e-6 * M * N * K / best:5.0f} TF/s best', flush=True)


for pre, M, d, h in (('enc', Me, 768, 3072), ('dec', Md, 512, 2048)):
    case(f'{pre} qkv  bf16 out', 'fwd', M, 3 * d, d, out16=True, bias=True)
    case(f'{pre} proj +res', 'fwd', M, d, d, out32=True, bias=True, res=True)
    case(f'{pre} fc1  GELU', 'fwd', M, h, d, out16=True, bias=True, epi=GELU | A16)
    case(f'{pre} fc2  +res', 'fwd', M, d, h, out32=True, bias=True, res=True)
    case(f'{pre} fc2 dgrad GELU\'+cs', 'dgrad', M, h, d, out16=True, epi=DGELU | A16, colsum=True)      # + the fc1 bias gradient as column sums (batch >= 8)
    case(f'{pre} fc2 dgrad GELU\'', 'dgrad', M, h, d, out16=True, epi=DGELU | A16 | CONSTS['VITAE_EPI_AUX_DERIV'])   # what the batch-4 step launches: the saved derivative, bias gradient by the wgrad half
    case(f'{pre} fc2 dgrad plain', 'dgrad', M, h, d, out16=True)
    case(f'{pre} fc1 dgrad', 'dgrad', M, d, h, out32=True)
    case(f'{pre} proj dgrad', 'dgrad', M, d, d, out32=True)
    case(f'{pre} qkv dgrad', 'dgrad', M, d, 3 * d, out32=True)
case('dec pred', 'fwd', Md, 16384, 512, out32=True, bias=True)
