"""Which bf16 rounding points of the benchmarked schedule carry its loss error?  

CPU experiment on the oracle (plain PyTorch fp32, oracle/mae_ref.py): the bf16 schedule of engine.py is EMULATED by rounding, class
by class, exactly the tensors the HIP path holds in bf16 — the operands of every dense contraction (activations and weights, forward;
output gradients, backward), q | k | v and their gradient, the attention probabilities, the saved fc1 pre-activation — while every
accumulation, the residual stream, LayerNorm, the loss chain and AdamW stay fp32 as they do on the GPU.  The pinned B = 4 trajectory
(tests/golden/vitb_b4.npz: three AdamW steps + the losses of a fourth batch, from the reference's own model) is then run with
    all      every class rounded (should land where the GPU's bf16 mode lands: total 1.5e-4, raw edge 6.6e-4, contrastive 8e-3)
    -X       every class but X          +X   only X
and the worst relative error of [total, raw edge, recon, contrastive] over the four steps is printed per variant.

    python tools/bf16_rounding_ablation.py [variant ...]      (no arguments: the default list)
Nothing here is imported by the product; the oracle is used as the model under test, the pins are the reference's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import mae_ref as R
from oracle import train_ref as T
from oracle.gen_golden import VITB

torch.set_num_threads(int(os.environ.get('ABL_THREADS', '8')))
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)

LIN_CLASSES = ['patch_embed', 'enc.qkv', 'enc.proj', 'enc.fc1', 'enc.fc2', 'decoder_embed', 'dec.qkv', 'dec.proj', 'dec.fc1', 'dec.fc2',
               'decoder_pred', 'predictor']
OTHER = ['enc.attn', 'dec.attn', 'hpre']          # q|k|v + P + dS in bf16; saved fc1 pre-activation in bf16
ALL = [c + ':fx' for c in LIN_CLASSES] + [c + ':fw' for c in LIN_CLASSES] + [c + ':b' for c in LIN_CLASSES] + OTHER       # forward activation / forward weight / backward operands
ON = set()


def cls_of(name):
    if name.startswith('blocks.'):
        return 'enc.' + name.split('.')[-2]
    if name.startswith('decoder_blocks.'):
        return 'dec.' + name.split('.')[-2]
    if name.startswith('predictor.'):
        return 'predictor'
    return name.rsplit('.', 1)[0].replace('.proj', '') if name.startswith('patch_embed') else name.rsplit('.', 1)[0]


class Lin(torch.autograd.Function):
    """y = x W^T (+ b) with the operand roundings of the LDS-DMA GEMMs: forward operands, backward dy / W / x."""
    @staticmethod
    def forward(ctx, x, w, b, rf, rb):
        xr, wr = (bf(x) if rf & 1 else x), (bf(w) if rf & 2 else w)
        ctx.save_for_backward(bf(x) if rb else x, bf(w) if rb else w)
        ctx.rb, ctx.has_b = rb, b is not None
        y = xr @ wr.t()
        return y + b if b is not None else y
    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyr = bf(dy) if ctx.rb else dy
        dx = dyr @ w
        dw = dyr.reshape(-1, dyr.shape[-1]).t() @ x.reshape(-1, x.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dw, db, None, None


class Rnd(torch.autograd.Function):
    """value rounded on the way forward and / or gradient rounded on the way back"""
    @staticmethod
    def forward(ctx, x, f, b):
        ctx.b = b
        return bf(x) if f else x
    @staticmethod
    def backward(ctx, g):
        return (bf(g) if ctx.b else g), None, None


class GeluSavedBf16(torch.autograd.Function):
    """exact GELU of the fp32 value; the backward evaluates GELU' at the bf16-rounded saved pre-activation (VITAE_EPI_AUX_BF16)"""
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(bf(h))
        return F.gelu(h)
    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        cdf = 0.5 * (1 + torch.erf(h * 0.7071067811865476))
        pdf = torch.exp(-0.5 * h * h) * 0.3989422804014327
        return g * (cdf + h * pdf)


NAMES = {}


def lin(x, w, b=None):
    c = NAMES.get(id(w))
    rf, rb = ((c + ':fx') in ON) | 2 * ((c + ':fw') in ON), (c + ':b') in ON
    if not rf and not rb:
        return F.linear(x, w, b)
    return Lin.apply(x, w, b, rf, rb)


def attention(x, sd, pre, heads):
    B, N, C = x.shape
    hd = C // heads
    a16 = ('enc.attn' if pre.startswith('blocks.') else 'dec.attn') in ON
    qkv = lin(x, sd[pre + 'qkv.weight'], sd[pre + 'qkv.bias'])
    qkv = Rnd.apply(qkv, a16, a16)                      # q | k | v and dqkv exist in bf16 only
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-2, -1)) * hd ** -0.5
    s = Rnd.apply(s, False, a16)                        # dS feeds the dq / dk MFMAs in bf16
    p = Rnd.apply(s.softmax(dim=-1), a16, False)        # P feeds the PV MFMA in bf16
    o = (p @ v).transpose(1, 2).reshape(B, N, C)
    o = Rnd.apply(o, False, a16)                        # dO is an MFMA operand of the backward
    return lin(o, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'])


def mlp(x, sd, pre):
    h = lin(x, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias'])
    h = GeluSavedBf16.apply(h) if 'hpre' in ON else F.gelu(h)
    return lin(h, sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias'])


def patch_embed(x, sd, p):
    w, b = sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias']
    B, C = x.shape[:2]
    g = [s // p for s in x.shape[2:]]
    # Conv3d(k = p, s = p) as the GEMM the HIP path runs: patches [B L, C p^3] x W[D, C p^3]^T
    pt = x.reshape(B, C, g[0], p, g[1], p, g[2], p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, g[0] * g[1] * g[2], C * p ** 3)
    NAMES[id(w)] = 'patch_embed'
    c = 'patch_embed'
    w2 = w.reshape(w.shape[0], -1)
    rf, rb = ((c + ':fx') in ON) | 2 * ((c + ':fw') in ON), (c + ':b') in ON
    return Lin.apply(pt, w2, b, rf, rb) if (rf or rb) else F.linear(pt, w2, b)


class _F:       # the functional namespace mae_ref sees: linear replaced, the rest untouched
    def __getattr__(self, k):
        return lin if k == 'linear' else getattr(F, k)


R.F = _F()
R._attention, R._mlp, R.patch_embed = attention, mlp, patch_embed


def run(variant):
    ON.clear()
    for tok in variant.split(','):
        tok = tok.strip()
        if tok in ('none', ''):
            continue
        if tok == 'all':
            ON.update(ALL)
        elif tok in ('fwdx', 'fwdw'):
            ON.update(c + (':fx' if tok == 'fwdx' else ':fw') for c in LIN_CLASSES)
        elif tok in ('fwd', 'bwd'):
            ON.update(c + sfx for c in LIN_CLASSES for sfx in ((':fx', ':fw') if tok == 'fwd' else (':b',)))
        elif tok[0] in '+-':
            sel = [a for a in ALL if a == tok[1:] or a.startswith(tok[1:] + ':') or a.startswith(tok[1:]) and tok[1:].endswith(':f') or (tok[1:].endswith('*') and a.startswith(tok[1:-1]))]
            assert sel, tok
            (ON.update if tok[0] == '+' else ON.difference_update)(sel)
        else:
            raise SystemExit(f'unknown token {tok}')
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'vitb_b4.npz'))
    B, steps, lr, wd, mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    B, steps = int(B), int(steps)
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    tr = T.RefTrainer(cfg, R.init_state_dict(cfg, seed=0), lr=lr, weight_decay=wd)
    NAMES.clear()
    for n, p in tr.params.items():
        if n.endswith('.weight') and p.ndim >= 2:
            NAMES[id(p)] = cls_of(n)
    worst, per = [0.0] * 4, []
    t0 = time.time()
    for it in range(steps + 1):
        v1, v2 = R.synthetic_views((4, 4, 96, 96, 96), seed=1234 + it)
        n1, n2 = R.masking_noise(4, cfg.num_patches, seed=4321 + it)
        terms, _, _ = tr.step(v1, v2, n1, n2, lr=lr, mask_ratio=mask_ratio, edge_map_weight=edge_w, contr_weight=contr_w)
        got = [terms['loss'] - terms['contr_loss'], terms['edge_map_loss'], terms['reconstruction_loss'], terms['contr_loss']]
        want = [g['losses'][it][i] for i in (0, 1, 2, 4)]
        e = [(a - b) / (abs(b) + 1e-12) for a, b in zip(got, want)]        # signed: the classes' contributions add with their signs
        per.append(e)
        worst = [max(a, abs(b)) for a, b in zip(worst, e)]
    print(f'{variant:28s} worst [total, edge, recon, contr] = ' + ' '.join(f'{w:9.2e}' for w in worst) +
          '   signed per step, total: ' + ' '.join(f'{p[0]:+8.1e}' for p in per) + '  edge: ' + ' '.join(f'{p[1]:+8.1e}' for p in per) + f'   ({time.time() - t0:.0f} s)', flush=True)


if __name__ == '__main__':
    variants = sys.argv[1:] or ['none', 'all', 'fwd', 'bwd', 'all,-decoder_pred', 'all,-dec.*,-decoder_pred,-decoder_embed', 'all,-enc.*,-patch_embed',
                                '+decoder_pred', '+dec.attn,+enc.attn', '+hpre']
    for v in variants:
        run(v)
