"""What the GELU' epilogue of the fc2 input gradient consists of (forced tile): plain bf16 out / + column sums / ReLU-mask (aux read, one
select) / GELU' / GELU' + column sums;  and the forward GELU: bf16 out only / GELU without ... (python tools/epi_ablate.py B=32 tile=0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib, CONSTS
from tools.bt_bench import graph_time
dev = 'cuda'
args = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
B, tile = int(args.get('B', 32)), int(args.get('tile', 0))
GELU, DGELU, RMASK, A16 = CONSTS['VITAE_EPI_GELU'], CONSTS['VITAE_EPI_DGELU'], CONSTS['VITAE_EPI_RELU_MASK'], CONSTS['VITAE_EPI_AUX_BF16']
P = lambda t: None if t is None else t.data_ptr()
for pre, M, d, h in (('enc', 2 * B * 55, 768, 3072), ('dec', B * 217, 512, 2048)):
    for form, (Mg, N, K) in (('dgrad', (M, h, d)), ('fwd', (M, h, d))):
        akc, bkc = (1, 0) if form == 'dgrad' else (1, 1)
        As = [torch.randn(Mg, K, device=dev).bfloat16() for _ in range(4)]
        Bs = [(torch.randn(N, K, device=dev) if bkc else torch.randn(K, N, device=dev)).bfloat16() for _ in range(4)]
        C16 = torch.empty(Mg, N, device=dev, dtype=torch.bfloat16)
        aux16 = torch.randn(Mg, N, device=dev).bfloat16()
        aux32 = torch.randn(Mg, N, device=dev)
        bias = torch.randn(N, device=dev)
        cs = torch.zeros(N, device=dev)
        ws = torch.zeros(1 << 24, device=dev)
        lib.vitae_gemm_glds_set_bt_tile(tile)
        split = lib.vitae_gemm_glds_pick_split_k(Mg, N, K)
        cnt = [0]
        def mk(epi, aux, colsum, b=None):
            def go():
                cnt[0] += 1; i = cnt[0] % 4
                lib.vitae_gemm_glds(akc, bkc, P(As[i]), K, P(Bs[i]), K if bkc else N, None, N, P(C16), N, Mg, N, K, P(b), None, N, epi, P(aux), N, 0,
                                    1 if (epi & 15) == GELU else split, P(ws), P(cs) if colsum else None, torch.cuda.current_stream().cuda_stream)
            return go
        if form == 'dgrad':
            cases = {'bf16 out': mk(0, None, False), 'bf16 out + colsum': mk(0, None, True), 'relu-mask(aux16)': mk(RMASK | A16, aux16, False),
                     "GELU'(aux16)": mk(DGELU | A16, aux16, False), "GELU'(aux16) + colsum": mk(DGELU | A16, aux16, True), "GELU'(aux32) + colsum": mk(DGELU, aux32, True)}
        else:
            cases = {'bf16 out + bias': mk(0, None, False, bias), 'GELU aux16': mk(GELU | A16, aux16, False, bias), 'GELU aux32': mk(GELU, aux32, False, bias)}
        print(f'--- {pre} {form} M={Mg} N={N} K={K} tile={tile} split={split}')
        for name, go in cases.items():
            go(); torch.cuda.synchronize()
            print(f'  {name:26s} {graph_time(go, 20):7.1f} us', flush=True)
lib.vitae_gemm_glds_set_bt_tile(-1)
