"""Host-side sequencing of the HIP hot path: one flat parameter arena, one activation workspace,
and hand-ordered forward / backward / optimiser launch lists on a single HIP stream.

This replaces what autograd + ATen do for the reference (``ContrastiveMAEViT.forward``,
model/vit_autoenc.py:270-285, driven by ``train_one_stage_epoch``, utils/train_one_epoch.py:21-110):
every arithmetic step is a kernel of ``libvitae_hip.so`` called through the C ABI
(``include/vitae_hip.h``); torch only owns the device memory and the stream.  Nothing here allocates
after ``_alloc`` or synchronises, so a whole optimisation step can be captured in a HIP graph
(``capture_train_step``) and replayed; lr / loss weights reach the kernels through the device-resident
``hp`` block.

Design points that differ from the reference on purpose (same numbers, less work):
  * the two encoder passes of the contrastive model (vit_autoenc.py:272,277) run as ONE pass over the
    2B concatenated samples (rows of every op are per-sample independent; BatchNorm of the predictor
    is still applied per view, :280-284);
  * the patch embedding is computed for the kept 25 % of the patches only (masking is decided first);
  * weight gradients of matrices are written with beta=0 on the first micro-step, so only the small
    vector/token segment of the gradient arena needs zeroing.
"""
from __future__ import annotations

import os

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ._abi import CONSTS, VitaeError, lib

_C = CONSTS
EPI_NONE, EPI_GELU, EPI_DGELU, EPI_RELU = (_C['VITAE_EPI_NONE'], _C['VITAE_EPI_GELU'], _C['VITAE_EPI_DGELU'],
                                           _C['VITAE_EPI_RELU_MASK'])
HP = {k[len('VITAE_HP_'):]: v for k, v in _C.items() if k.startswith('VITAE_HP_')}
PREC = {'fp32': _C['VITAE_PREC_F32'], 'bf16': _C['VITAE_PREC_BF16'], 'fp32x3': _C['VITAE_PREC_BF16X3']}


def _triple(v):
    if isinstance(v, (tuple, list)):
        assert len(v) == 3
        return tuple(int(i) for i in v)
    return (int(v),) * 3


@dataclass
class MAEConfig:
    """Constructor arguments of the reference model classes (model/vit_autoenc.py:18-21,242-245)."""
    volume_size: Tuple[int, int, int] = (96, 96, 96)
    patch_size: int = 16
    in_chans: int = 4
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    contrastive: bool = False
    ln_eps: float = 1e-6

    def __post_init__(self):
        self.volume_size = _triple(self.volume_size)
        self.patch_size = int(self.patch_size if not isinstance(self.patch_size, (tuple, list)) else self.patch_size[0])

    @property
    def grid(self):
        p = self.patch_size
        return tuple(v // p for v in self.volume_size)

    @property
    def num_patches(self):
        g = self.grid
        return g[0] * g[1] * g[2]

    @property
    def patch_dim(self):
        return self.patch_size ** 3 * self.in_chans

    def len_keep(self, mask_ratio):
        return int(self.num_patches * (1 - mask_ratio))


def gaussian_taps_host(sigma: float = 2.0) -> np.ndarray:
    """1-D taps of model/model_utils/gaussian_filter.py:5-13 (ks = int(5*sigma) made odd, samples at
    linspace(-ks//2, ks//2+1, ks)), normalised in float64 so that k (x) k (x) k equals the reference's
    renormalised dense kernel (gaussian_filter.py:22-23) to fp32 round-off."""
    ks = int(sigma * 5)
    if ks % 2 == 0:
        ks += 1
    ts = np.linspace(float(-ks // 2), float(ks // 2 + 1), ks, dtype=np.float64)
    g = np.exp(-(ts / sigma) ** 2 / 2)
    return (g / g.sum()).astype(np.float32)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_ABLATE_LN_FWD = os.environ.get('VITAE_ABLATE_LN_FWD') == '1'


class HipMAEEngine:
    """Owns arenas + workspace for one model instance on one GPU and sequences the kernels."""

    def __init__(self, cfg: MAEConfig, params: "OrderedDict[str, torch.Tensor]",
                 buffers: Dict[str, torch.Tensor], device: torch.device, precision: str = 'fp32'):
        if device.type != 'cuda':
            raise VitaeError('HipMAEEngine needs a ROCm device: the hot path is HIP-only (no CPU fallback)')
        lib.load()
        self.cfg, self.device = cfg, device
        self.prec = PREC[precision]
        self.precision = precision
        D, Dd = cfg.embed_dim, cfg.decoder_embed_dim
        if D % cfg.num_heads or Dd % cfg.decoder_num_heads:
            raise VitaeError('embed dims must be divisible by the head counts')
        self.hd, self.hdd = D // cfg.num_heads, Dd // cfg.decoder_num_heads
        self.Hm, self.Hmd = int(D * cfg.mlp_ratio), int(Dd * cfg.mlp_ratio)
        if cfg.patch_size % 4:
            raise VitaeError('patch_size must be a multiple of 4 (16-byte gathers)')
        self._build_arena(params)
        # bf16 activations + LDS-DMA GEMMs for the transformer blocks, decoder_pred and the patch embedding:
        # needs every contraction length to be a multiple of 64 and an MFMA attention head size
        dims = (D, Dd, self.Hm, self.Hmd, cfg.patch_dim)
        self.act16 = (self.prec == PREC['bf16'] and all(v % 64 == 0 for v in dims)
                      and self.hd in (32, 64) and self.hdd in (32, 64))
        self.wgrad_side = os.environ.get('VITAE_WGRAD_SIDE', '1') != '0'
        self.target_fork = os.environ.get('VITAE_TARGET_FORK', 'start')
        # What fc1 saves for the fc2 input gradient is bf16 (VITAE_EPI_AUX_BF16) in bf16 mode, at EVERY size (default on since the end of
        # round 4: batch 4 4.50 -> 4.43 ms, batch 32 10.66 -> 10.58, patch 8 10.90 -> 10.84 once the big-tile epilogue took the aux type
        # as a template parameter; VITAE_HPRE_BF16=0 keeps fp32).  Two documented bf16-mode deviations ride on it: the backward's GELU'
        # comes from a bf16 value (since round 5: GELU'(fp32 pre-activation) rounded once, VITAE_EPI_AUX_DERIV), and decoder_pred's
        # bias gradient is the column sum of the bf16 loss gradient.  The bf16 test bounds (tests/test_gpu_model.py) were recorded
        # with both in place.
        self._hpre16_env = os.environ.get('VITAE_HPRE_BF16', 'auto')
        self.hpre16 = False
        self._aux16 = 0
        # What fc1's epilogue saves for the backward (round 5): GELU'(pre-activation) instead of the pre-activation itself
        # (VITAE_EPI_AUX_DERIV) — the gate that gives GELU gives GELU' for one more FMA, and the fc2 input-gradient epilogue becomes
        # a multiply: its ~20 VALU operations per element were 11 of the 46 us of that launch at batch 32 (tools/epi_ablate.py).
        # fp32 aux: the same value bit for bit; bf16 aux: GELU' of the fp32 pre-activation, rounded once (was GELU' of the rounded one).
        self._auxd = _C['VITAE_EPI_AUX_DERIV'] if os.environ.get('VITAE_AUX_DERIV', '1') != '0' else 0
        self.fc1_bias_by_wgrad = os.environ.get('VITAE_FC1_BIAS_BY_WGRAD', '1') != '0'   # (see _block_bwd16)
        self._init_w2()
        # the gradient norm's matrix share is accumulated by the weight-gradient epilogues themselves (vitae_gemm_glds_set_wgrad_sqnorm)
        # instead of a pass over each bucket (45 us per bucket, the last one exposed behind the backward); single process only — a
        # data-parallel norm is the norm of the REDUCED gradients
        self.epi_norm = self.act16 and os.environ.get('VITAE_EPI_GRADNORM', '1') != '0'
        # q | k | v leave the qkv GEMM in bf16 only and the attention kernels read that (no fp32 qkv in HBM: the GEMM epilogue is
        # bound by its output bytes, and the kernels no longer convert while staging); needs the one-launch attention backward
        self.qkv16 = self.act16 and os.environ.get('VITAE_QKV_BF16', '1') != '0'
        self.buffers = buffers   # pos_embed, decoder_pos_embed, BN running stats (device tensors)
        f32 = dict(dtype=torch.float32, device=device)
        # ring of pinned staging buffers: the host may run a few steps ahead of the stream, so the
        # buffer an in-flight async copy reads from must not be rewritten by the next step
        self.hp_vals = [0.0] * _C['VITAE_HP_COUNT']
        # Host -> device channel of the per-step scalars (lr, betas, loss multipliers, "keep the injected masking noise"): a
        # pinned ring of 16 blocks.  The fused step fetches block (step_seq mod 16) with its first launch (vitae_step_prologue,
        # inside the captured graph) — step_seq lives on the device and is advanced by the step's last launch, the host counts
        # the steps it has launched (step_seq_host): no host-to-device copy sits between two graph replays.  Everything else
        # that reads hp (generic autograd route, stand-alone forward) gets it by ``flush_hparams`` (an ordinary copy).
        self.hp_ring = torch.zeros(16, _C['VITAE_HP_COUNT'], dtype=torch.float32).pin_memory()
        # staging ring of ``flush_hparams`` (launches outside the fused step, where step_seq_host does not advance): every flush
        # takes the NEXT slot, so a copy still in flight is never rewritten by the following set_hparams / flush
        self.hp_stage = torch.zeros(16, _C['VITAE_HP_COUNT'], dtype=torch.float32).pin_memory()
        self._hp_stage_i = 0
        self.step_seq_host = 0
        self.step_seq = torch.zeros(1, dtype=torch.int64, device=device)
        self._hp_dirty = True
        # Masking noise of the fused step (vit_autoenc.py:139's torch.rand): Philox4x32-10 keyed by (noise_seed, step_seq) inside the
        # step's first launch — a DIFFERENT random stream from torch's generator (documented deviation; injected noise reproduces
        # the reference bit for bit).  The seed follows torch's: ``_noise_seed_now`` re-derives it from torch.initial_seed() and the
        # data-parallel rank whenever the user has reseeded (a manual_seed after the first step takes effect at the next capture /
        # eager launch; ranks that did not seed with seed + rank still draw different masks).
        self._seed_src = None
        self.noise_seed = 0
        self._noise_seed_now()
        self.hp = torch.zeros(_C['VITAE_HP_COUNT'], **f32)
        self.acc = torch.zeros(_C['VITAE_ACC_COUNT'], dtype=torch.float64, device=device)
        self.losses = torch.zeros(8, **f32)   # [loss, raw_edge, recon, percep, contr, grad_norm, -, -]
        self.taps = gaussian_taps_host(2.0)
        self._taps_c = self.taps.ctypes.data
        # fp32x3: the GEMMs of the fp32 schedule on the wave-specialised kernel whose producer waves split the fp32 operands
        # (vitae_gemm_wsx3: in-launch split-K — its workspaces start with zeroed tickets — and bias gradients beside the weight gradients)
        self.x3ws = self.prec == PREC['fp32x3'] and os.environ.get('VITAE_X3_WS', '1') != '0'
        # fp32x3: the weight-gradient launches of vitae_gemm_wsx3 add their squares to the norm as well; WHICH matrices they serve depends
        # on the shapes (_x3_ok), so the covered ranges are recorded as the launches are enqueued (_x3_cov) and a bucket reads the rest
        if self.x3ws and os.environ.get('VITAE_EPI_GRADNORM', '1') != '0':
            self.epi_norm = True
        self._x3_cov = []
        self.ws = torch.empty(1 << 24, **f32)   # split-K scratch (64 MiB)
        # (tickets + partial tiles of vitae_gemm_wsx3, one per stream; NOT shared with the older kernels, which park plain partials at offset 0)
        self.ws_x3 = {k: torch.zeros(n, **f32) for k, n in (('main', 1 << 23), ('pside', 1 << 22), ('wside', 1 << 22))} if self.x3ws else None
        self.ws16 = torch.zeros(1 << 24, **f32)  # LDS-DMA GEMMs: tile tickets (kept zero by the kernels) + partial tiles
        self.attn_bias_colsum = os.environ.get('VITAE_ATTN_BIAS_COLSUM', '1') != '0'
        # predictor Linears on bf16 operands (LDS-DMA GEMMs) instead of the fp32-activation kernel (round 4: 6 launches of 70-110 us on
        # the predictor branch at batch 32, beside a chain that has no spare CUs there)
        self.pred16 = self.act16 and (cfg.embed_dim % 64 == 0) and os.environ.get('VITAE_PREDICTOR_BF16', '1') != '0'
        self.ws16_side = torch.zeros(1 << 22, **f32) if self.pred16 else None
        # grouped weight gradients of a block (many token rows) on the wgrad side stream: they fill the CUs the dependent chain's
        # 168-220-tile launches leave idle; two alternating sets of dy operands, a set is rewritten two blocks later
        self.wgrad_group_side = os.environ.get('VITAE_WGRAD_GROUP_SIDE', '1') == '1'
        self.ws16_wside = torch.zeros(1 << 24, **f32) if (self.act16 and self.wgrad_group_side) else None
        self.ln_part_on = os.environ.get('VITAE_LN_PART', '1') != '0'
        # LayerNorm backward through partial records (no atomics) from this many elements on.  Round 6: at EVERY size — the records of all
        # LayerNorms of a step are summed by ONE launch in front of the optimiser's tail (``ln_flush_once``), not by one per backward
        # phase on the dependent chain (that form lost to the 3 D float atomics per workgroup at batch 4: 10.8 + 4 x 13 us against 10.2)
        self.ln_part_min = int(float(os.environ.get('VITAE_LN_PART_MIN', '0')))
        self.ln_flush_once = os.environ.get('VITAE_LN_FLUSH_ONCE', '1') != '0'
        self._ln_pending = []
        self.B = None
        self.buf: Dict[str, torch.Tensor] = {}
        # one workspace per (batch, kept patches), kept alive while captured graphs may hold its addresses; evicting one
        # bumps ``ws_gen`` and every step runner drops its graphs (model/vit_autoenc.py: _StepRunner.run)
        self._ws_cache: "OrderedDict[Tuple[int, int], dict]" = OrderedDict()
        self.ws_gen = 0
        self.opt_state = None
        self.opt_step = 0
        self._accum = False
        self._split_cache: Dict[Tuple[int, int, int], int] = {}
        self.stream = 0
        # Branch streams (properties ``side`` / ``wside`` / ``pside`` / ``oside`` below):
        #   side   target-side loss branch beside the transformer        wside  weight-gradient GEMMs beside the dgrad chain
        #   pside  predictor branch (fwd, cosine loss, bwd) beside the decoder    oside  per-bucket grad-norm + AdamW beside the backward
        self._branch_streams = {n: torch.cuda.Stream(device=device) for n in ('side', 'wside', 'pside', 'oside')}
        sel = os.environ.get('VITAE_SIDE_STREAMS', 'auto')       # 'auto' | 'all' | 'none' | comma list of the four names
        self.side_streams = None if sel == 'auto' else set(self._branch_streams) if sel == 'all' else set(filter(None, sel.split(','))) - {'none'}
        self.side_min_rows = int(os.environ.get('VITAE_SIDE_MIN_ROWS', '2500'))
        self.ws_side = torch.empty(1 << 22, **f32)     # its own split-K scratch
        self.ws_wside = torch.empty(1 << 23, **f32)    # ... and the wgrad side stream's (fp32 / fp32x3 schedules: split weight gradients)
        self.overlap_predictor = os.environ.get('VITAE_PREDICTOR_SIDE', '1') != '0'
        self.overlap_optimizer = os.environ.get('VITAE_OPT_IN_BACKWARD', '1') != '0'
        self._opt_pending = False
        self._pred_pending = False
        self._pred_joined = False   # the predictor branch was joined by backward_dec(part='top')
        self.overlap_wgrad = True
        self._wg_events: Dict[str, torch.cuda.Event] = {}
        self._wg_pending = set()
        self._wire_ready = False
        self.grads_wire16 = None  # set by the data-parallel reducer when gradients are exchanged (and consumed) in bf16
        self.gemm_timer = None   # bench.py: list collecting (start_event, end_event, flops, kernel tag, scope) per launch
        self._scope = None       # 'enc' / 'dec' while the launches of a transformer block of that stack are issued
        self.set_hparams(lr=0.0, beta1=0.9, beta2=0.95, eps=1e-8, bc1=1.0, bc2=1.0, grad_mul=1.0, g_recon=1.0,
                         g_edge=0.0, g_contr=0.0, edge_w=0.0, contr_w=0.0)

    # ------------------------------------------------------------------ arenas
    def _build_arena(self, params: "OrderedDict[str, torch.Tensor]"):
        """Flat fp32 arena: [matrices | cls/mask tokens | vectors]; every tensor 16-byte aligned.
        AdamW decays the first two segments (timm add_weight_decay: no decay iff ndim == 1 or
        name.endswith('.bias')); the last two are the atomically-accumulated gradients that are
        zeroed every step."""
        mats, toks, vecs = [], [], []
        for n, p in params.items():
            if p.ndim <= 1 or n.endswith('.bias'):
                vecs.append(n)
            elif n in ('cls_token', 'mask_token'):
                toks.append(n)
            else:
                mats.append(n)
        self.layout: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        off = 0
        for n in mats + toks + vecs:
            if n == (toks[0] if toks else None):
                self.tok_off = off
            if n == (vecs[0] if vecs else None):
                self.vec_off = off
            shp = tuple(params[n].shape)
            self.layout[n] = (off, shp)
            off += (int(np.prod(shp)) + 3) // 4 * 4
        if not toks:
            self.tok_off = self.vec_off
        self.n_total = off
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for n, (o, shp) in self.layout.items():
            k = int(np.prod(shp))
            self.p[n] = self.params[o:o + k].view(shp)
            self.g[n] = self.grads[o:o + k].view(shp)
            self.p[n].copy_(params[n].detach().to(device=dev, dtype=torch.float32))
        # bf16 weight shadow for the throughput-mode GEMMs (kept current by the fused AdamW kernel;
        # any other in-place change of the fp32 arena bumps its torch version counter -> re-cast)
        self.params16, self.p16, self._shadow_version = None, {}, None
        if self.prec == PREC['bf16']:
            self.params16 = torch.zeros(off, dtype=torch.bfloat16, device=dev)
            for n, (o, shp) in self.layout.items():
                self.p16[n] = self.params16[o:o + int(np.prod(shp))].view(shp)

    # ------------------------------------------------------------------ two-plane weights (bf16 mode)
    def _init_w2(self):
        """bf16 mode: the forward of the decoder's fc1 multiplies by hi + lo planes of its weight (vitae_gemm_glds_w2).  Why that layer:
        tools/bf16_rounding_ablation.py (CPU emulation of this schedule on the oracle, pinned B = 4 trajectory of
        tests/golden/vitb_b4.npz) — the loss error of the bf16 schedule is the rounding of the WEIGHTS in the forward (total 1.3e-4 of
        1.5e-4; activations 9e-6, the whole backward 4e-6), the decoder's fc1 alone +1.2e-4 (raw edge +4.9e-4); with its weight at
        2^-17 the emulation lands at 6e-5 / 2.2e-4.  The planes ([W hi | W lo] side by side, 8 x 2 M bf16) are rewritten behind the AdamW
        launch that covers them.  VITAE_W2='' switches it off, VITAE_W2='dec.fc1,dec.fc2' names more classes of the decoder blocks."""
        self._w2 = {}          # weight name -> (lo-plane tensor view)
        self._w2_groups = []   # (first arena offset, tensor length, stride, count, lo tensor [count, len])
        cls = [c for c in os.environ.get('VITAE_W2', 'dec.fc1').split(',') if c]
        if not (self.act16 and cls):
            return
        if 'decoder' in cls:       # shorthand: every Linear of the decoder (round 6: the raw edge term needs ALL of them, LABNOTES)
            cls = [c for c in cls if c != 'decoder'] + ['dec.qkv', 'dec.proj', 'dec.fc1', 'dec.fc2', 'decoder_pred', 'decoder_embed']
        for c in dict.fromkeys(cls):
            if c in ('decoder_pred', 'decoder_embed'):      # a Linear outside the block stacks: a group of one
                names, depth = [c + '.weight'], 1
            else:
                stack, leaf = c.split('.')
                pre = {'dec': 'decoder_blocks', 'enc': 'blocks'}[stack]
                depth = self.cfg.decoder_depth if stack == 'dec' else self.cfg.depth
                sub = {'fc1': 'mlp.fc1', 'fc2': 'mlp.fc2', 'qkv': 'attn.qkv', 'proj': 'attn.proj'}[leaf]
                names = [f'{pre}.{i}.{sub}.weight' for i in range(depth)]
            offs = [self.layout[n][0] for n in names]
            ln = int(np.prod(self.layout[names[0]][1]))
            stride = offs[1] - offs[0] if depth > 1 else ln
            kdim = int(self.layout[names[0]][1][1])
            if any(offs[i + 1] - offs[i] != stride for i in range(depth - 1)) or ln % 4 or stride % 4 or kdim % 64 or kdim < 128:
                continue       # (not equally spaced in this arena, or a reduction of fewer than two k-tiles: the class keeps its one-plane forward)
            rows = int(self.layout[names[0]][1][0])
            hilo = torch.zeros(depth, rows, 2 * kdim, dtype=torch.bfloat16, device=self.device)     # [W hi | W lo] side by side
            self._w2_groups.append((offs[0], ln, stride, depth, hilo, rows, kdim))
            for i, n in enumerate(names):
                self._w2[n] = hilo[i]

    def refresh_w2(self, stream=None, lo=0, hi=None):
        """Rewrite the lo planes of the two-plane weights that lie inside arena range [lo, hi) (default: all of them)."""
        hi = self.n_total if hi is None else hi
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        for off0, ln, stride, count, t, rows, kdim in self._w2_groups:
            inside = [i for i in range(count) if off0 + i * stride >= lo and off0 + i * stride + ln <= hi]
            # The two-plane forward reads BOTH planes from ``hilo`` (never the shadow), so a tensor that [lo, hi) only touches would
            # keep a stale hi plane with no error anywhere: every two-plane tensor overlapping the range must lie inside it, and the
            # hits must be one contiguous run (they are one strided cast launch).  Holds for ddp.engine_bucket_ranges' cuts, which
            # fall on the first matrix of a block; a future cut through a block fails HERE.
            touched = [i for i in range(count) if off0 + i * stride < hi and off0 + i * stride + ln > lo]
            if touched != inside or (inside and inside != list(range(inside[0], inside[0] + len(inside)))):
                raise VitaeError(f'refresh_w2: arena range [{lo}, {hi}) cuts through a two-plane weight (tensors {touched} of the group at '
                                 f'{off0}, fully inside: {inside}); optimiser buckets must end on tensor boundaries of these weights')
            if not inside:
                continue
            i0, n = inside[0], len(inside)          # (a bucket covers a contiguous run of blocks)
            lib.vitae_cast_bf16_hilo(self.params.data_ptr() + 4 * (off0 + i0 * stride), t[i0].data_ptr(), rows, kdim, stride, n, st)

    def refresh_shadow(self, force: bool = False):
        if self.params16 is None:
            return
        if force or self._shadow_version != self.params._version:
            lib.vitae_cast_bf16(self.params.data_ptr(), self.params16.data_ptr(), self.n_total,
                                torch.cuda.current_stream(self.device).cuda_stream)
            self.refresh_w2()
            self._shadow_version = self.params._version

    # ------------------------------------------------------------------ hyper-parameters
    def set_hparams(self, **kw):
        """Host values of hp[0 .. VITAE_HP_HOST_COUNT): published in the ring block of the NEXT step to be launched."""
        for k, v in kw.items():
            self.hp_vals[HP[k.upper()]] = float(v)
        self.hp_ring[self.step_seq_host % self.hp_ring.shape[0]].copy_(torch.tensor(self.hp_vals, dtype=torch.float32))
        self._hp_dirty = True

    def flush_hparams(self):
        """For launches outside the fused step: the host slots of the device block by an ordinary (stream-ordered) copy."""
        if self._hp_dirty:
            n = _C['VITAE_HP_HOST_COUNT']
            self._hp_stage_i = (self._hp_stage_i + 1) % self.hp_stage.shape[0]
            stage = self.hp_stage[self._hp_stage_i]
            stage.copy_(torch.tensor(self.hp_vals, dtype=torch.float32))
            self.hp[:n].copy_(stage[:n], non_blocking=True)
            self._hp_dirty = False

    def _rank_now(self) -> int:
        """Data-parallel rank of this process: torch.distributed when it is initialised, else an explicit ``noise_rank`` (set by
        ``enable_data_parallel`` on the native RCCL route, which may run without a torch process group)."""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return int(dist.get_rank())
        except Exception:
            pass
        return int(getattr(self, 'noise_rank', 0))

    def _noise_seed_now(self) -> int:
        """Philox key of the masking noise: re-derived whenever torch's seed OR the rank changes (a process group initialised after
        the engine was built must still give every rank its own masks); a graph captured earlier keeps the key it was
        captured with, so ``enable_data_parallel`` drops the captured graphs."""
        src = (int(torch.initial_seed()), self._rank_now())
        if src != self._seed_src:
            self._seed_src = src
            self.noise_seed = (src[0] * 0x9E3779B97F4A7C15 + src[1] * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & ((1 << 63) - 1)
        return self.noise_seed

    def step_prologue(self, noise: torch.Tensor, accumulate: bool):
        """First launch of a fused step (captured with it): hp <- ring, masking noise, acc <- 0, token / vector gradients <- 0."""
        self._accum = bool(accumulate)
        self._x3_cov = []          # (a new grad window, like begin_grad_window: the fp32x3 coverage list restarts with every fused step)
        st = torch.cuda.current_stream(self.device).cuda_stream
        n = self.n_total - self.tok_off
        lib.vitae_step_prologue(self.hp.data_ptr(), self.hp_ring.data_ptr(), self.hp_ring.shape[0], self.step_seq.data_ptr(),
                                noise.data_ptr(), noise.numel(), self._noise_seed_now(), self.acc.data_ptr(),
                                None if accumulate else self.grads.data_ptr() + self.tok_off * 4, 0 if accumulate else n * 4, st)
        self._hp_dirty = False
        self._head_done = True

    def step_epilogue(self):
        """Last launch of a fused step: the device's step sequence number moves on (the host's: ``end_step_host``)."""
        lib.vitae_step_epilogue(self.step_seq.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)

    def end_step_host(self):
        """Called once per LAUNCHED fused step (graph replay or eager): the next step reads the next ring block, which starts out
        as a copy of the current values."""
        self.step_seq_host += 1
        self.hp_ring[self.step_seq_host % self.hp_ring.shape[0]].copy_(torch.tensor(self.hp_vals, dtype=torch.float32))

    _head_done = False

    def read_opt_step(self) -> int:
        """Number of AdamW steps actually applied (device-side count: a step skipped for a non-finite gradient norm is not in it)."""
        if self.opt_state is None:
            return 0
        self.opt_step = int(round(float(self.hp[_C['VITAE_HP_STEP']].item())))
        return self.opt_step

    def write_opt_step(self, step: int):
        self.opt_step = int(step)
        self.hp[_C['VITAE_HP_STEP']] = float(step)

    # ------------------------------------------------------------------ workspace
    _WS_ATTRS = ('B', 'keep', 'Be', 'Ne', 'Nd', 'Me', 'Md', 'Mpe', 'Mpd', 'Mpt', 'Mpl', 'R', 'mask_sum', 'edge_count', 'buf', 'hpre16', '_aux16')
    _WS_MAX = int(os.environ.get('VITAE_WORKSPACES', '4'))

    def _alloc(self, B: int, mask_ratio: float):
        cfg = self.cfg
        keep = cfg.len_keep(mask_ratio)
        if keep <= 0 or keep >= cfg.num_patches:
            raise VitaeError(f'mask_ratio {mask_ratio} leaves {keep} of {cfg.num_patches} patches')
        if self.B == B and self.keep == keep:
            return
        if self.B is not None:      # park the current workspace: graphs captured on it stay valid
            self._ws_cache[(self.B, self.keep)] = {k: getattr(self, k) for k in self._WS_ATTRS if hasattr(self, k)}
            self._ws_cache.move_to_end((self.B, self.keep))
        hit = self._ws_cache.get((B, keep))
        if hit is not None:
            for k, v in hit.items():
                setattr(self, k, v)
            return
        while len(self._ws_cache) >= self._WS_MAX:
            self._ws_cache.popitem(last=False)      # its buffers go back to the allocator: graphs holding them are stale
            self.ws_gen += 1
        self.B, self.keep = B, keep
        self.Be = 2 * B if cfg.contrastive else B
        L, P, D, Dd = cfg.num_patches, cfg.patch_dim, cfg.embed_dim, cfg.decoder_embed_dim
        Ne, Nd, Be = keep + 1, L + 1, self.Be
        Me, Md = Be * Ne, B * Nd
        self.Ne, self.Nd, self.Me, self.Md = Ne, Nd, Me, Md
        self.hpre16 = self.act16 and self._hpre16_env != '0'      # (end of round 4: a gain at every size — batch 32 10.66 -> 10.58 ms, patch 8 10.90 -> 10.84)
        self._aux16 = (CONSTS['VITAE_EPI_AUX_BF16'] if self.hpre16 else 0) | self._auxd
        dev = self.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        b = self.buf = {}
        V = cfg.volume_size[0] * cfg.volume_size[1] * cfg.volume_size[2]
        b['noise'] = f(Be, L)
        b['ids_shuffle'] = torch.empty(Be, L, dtype=torch.int32, device=dev)
        b['ids_restore'] = torch.empty(Be, L, dtype=torch.int32, device=dev)
        b['ids_restore64'] = torch.empty(Be, L, dtype=torch.int64, device=dev)
        b['mask'] = f(Be, L)
        b['patches'] = f(Be * keep, P)
        b['tok'] = f(Be * keep, D)
        b['dtok'] = f(Be * keep, D)

        def stack(pre, depth, M, d, h, N, hd):
            q32 = not (self.act16 and self._qkv16_ok(N, hd))   # fp32 q | k | v and their gradient exist only when the bf16 copy is not the operand
            b[pre + 'x'] = [f(M, d) for _ in range(depth + 1)]
            for i in range(depth):
                q = f'{pre}{i}.'
                b[q + 'y1'], b[q + 'mean1'], b[q + 'rstd1'] = f(M, d), f(M), f(M)
                b[q + 'qkv'], b[q + 'o'] = (f(M, 3 * d) if q32 else None), f(M, d)
                b[q + 'xmid'], b[q + 'y2'], b[q + 'mean2'], b[q + 'rstd2'] = f(M, d), f(M, d), f(M), f(M)
                b[q + 'hpre'] = torch.empty(M, h, dtype=torch.bfloat16, device=dev) if self.hpre16 else f(M, h)
                b[q + 'act'] = f(M, h)
            b[pre + 'dx'], b[pre + 'dy'], b[pre + 'do'] = f(M, d), f(M, d), f(M, d)
            b[pre + 'dh'], b[pre + 'dqkv'] = f(M, h), (f(M, 3 * d) if q32 else None)

        stack('enc', cfg.depth, Me, D, self.Hm, self.Ne, self.hd)
        stack('dec', cfg.decoder_depth, Md, Dd, self.Hmd, self.Nd, self.hdd)
        if self.act16:
            # bf16 GEMM operands, token rows padded to 64 with zeros (wgrad reduces over the padded count)
            z16 = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
            pad = lambda m: (m + 63) // 64 * 64
            self.Mpe, self.Mpd, self.Mpt = pad(Me), pad(Md), pad(Be * keep)
            for pre, depth, Mp, d, h in (('enc', cfg.depth, self.Mpe, D, self.Hm), ('dec', cfg.decoder_depth, self.Mpd, Dd, self.Hmd)):
                for i in range(depth):
                    q = f'{pre}{i}.'
                    b[q + 'y1_16'], b[q + 'o_16'], b[q + 'y2_16'], b[q + 'act_16'] = z16(Mp, d), z16(Mp, d), z16(Mp, d), z16(Mp, h)
                    if self.qkv16:
                        b[q + 'qkv_16'] = z16(Mp, 3 * d)
                b[pre + 'dx_16'], b[pre + 'dh_16'], b[pre + 'dqkv_16'] = z16(Mp, d), z16(Mp, h), z16(Mp, 3 * d)
                if self._grouped(Mp, d):
                    b[pre + 'dx_16b'] = z16(Mp, d)
                    if self.wgrad_group_side:
                        for nm, w in (('dx_16', d), ('dx_16b', d), ('dh_16', h), ('dqkv_16', 3 * d)):
                            b[pre + nm + '_alt'] = z16(Mp, w)
            b['dn_16'] = z16(self.Mpd, Dd)
            b['dpred_16'] = z16(self.Mpd, P)
            b['patches_16'], b['dtok_16'] = z16(self.Mpt, P), z16(self.Mpt, D)
            # decoder_embed on the LDS-DMA GEMM: its input (the view-1 rows of the latent) and its output gradient in bf16.
            # The wgrad reduces over pad(B * Ne) rows: the pad rows of de_16 stay zero, so whatever finite rows of latent_16
            # (view 2) sit opposite them contribute nothing.
            self.Mpl = pad(B * Ne)
            b['latent_16'], b['de_16'] = z16(max(self.Mpe, self.Mpl), D), z16(self.Mpl, Dd)
            if cfg.contrastive and self.pred16:
                # predictor on the LDS-DMA GEMMs: bf16 copies of its activations / their gradients (2 R = Me rows, zero pad rows)
                b['pr_16'], b['dp_16'], b['dph_16'] = z16(self.Mpe, D), z16(self.Mpe, D), z16(self.Mpe, D)
        if self.ln_part_on and max(Me * D, Md * Dd) >= self.ln_part_min:
            # LayerNorm backward: one record of column partials [d gamma | d beta | colsum(dx)] per workgroup and LayerNorm instance
            for pre_, depth, M_, d_ in (('blocks.', cfg.depth, Me, D), ('decoder_blocks.', cfg.decoder_depth, Md, Dd)):
                G = lib.vitae_layernorm_bwd_part_records(M_)
                for i in range(depth):
                    b[f'lnpart.{pre_}{i}.norm1.'], b[f'lnpart.{pre_}{i}.norm2.'] = f(G * 3 * d_), f(G * 3 * d_)
            b['lnpart.norm.'] = f(lib.vitae_layernorm_bwd_part_records(Me) * 3 * D)
            b['lnpart.decoder_norm.'] = f(lib.vitae_layernorm_bwd_part_records(Md) * 3 * Dd)
        for i in range(cfg.depth):
            b[f'enc{i}.lse'] = f(Be * cfg.num_heads * Ne)
        for i in range(cfg.decoder_depth):
            b[f'dec{i}.lse'] = f(B * cfg.decoder_num_heads * Nd)
        b['delta'] = f(max(Be * cfg.num_heads * Ne, B * cfg.decoder_num_heads * Nd))
        b['latent'], b['lat_mean'], b['lat_rstd'], b['dlatent'] = f(Me, D), f(Me), f(Me), f(Me, D)
        b['e'], b['de'] = f(B * Ne, Dd), f(B * Ne, Dd)
        b['dn'], b['dn_mean'], b['dn_rstd'], b['ddn'] = f(Md, Dd), f(Md), f(Md), f(Md, Dd)
        b['predfull'] = f(B, Nd, P)
        b['dpredfull'] = torch.zeros(B, Nd, P, dtype=torch.float32, device=dev)   # cls rows stay zero
        b['pred_vol'] = f(B, cfg.in_chans, *cfg.volume_size)
        b['blur_tmp'], b['blurred'] = f(B * cfg.in_chans * V), f(B * cfg.in_chans * V)
        b['edge_t'], b['edge_p'] = f(B * V), f(B * V)
        if cfg.in_chans not in (1, 4):   # the one-pass loss backward covers C = 1 and 4; others need the dG scratch
            b['dG'] = f(B * cfg.in_chans * 3 * V)
        if cfg.contrastive:
            R = B * Ne
            self.R = R
            b['ph'], b['pr'], b['pout'] = f(2 * R, D), f(2 * R, D), f(2 * R, D)
            b['bn_mean'], b['bn_rstd'] = f(2, D), f(2, D)
            # many rows: BatchNorm with the rows split over workgroups (two launches per op; csrc/norm.hip) — one 64-column strip
            # per workgroup is 12 workgroups at D = 768 whatever R is (batch 32 / patch 8: 45 + 78 us per view)
            if R >= self.bn_split_min_rows:
                b['bn_ws'] = f(int(lib.vitae_bn1d_split_ws_floats(R, D)))
            b['dp'], b['dpr'], b['dph'] = f(2 * R, D), f(2 * R, D), f(2 * R, D)
        self.mask_sum = float(B * (L - keep))
        self.edge_count = B * V

    # ------------------------------------------------------------------ predictor branch beside the decoder
    class _OnPredictorStream:
        """Launch helpers issue on ``pside`` with the branch's own split-K scratch inside this context."""

        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            e = self.eng
            self.saved = (e.stream, e.ws, e.ws16)
            self.ctx = torch.cuda.stream(e.pside)
            self.ctx.__enter__()
            e.stream, e.ws = e.pside.cuda_stream, e.ws_side
            if e.ws16_side is not None:
                e.ws16 = e.ws16_side        # split-K tickets / partials of the branch's LDS-DMA GEMMs: never shared with the chain's
            return e

        def __exit__(self, *a):
            e = self.eng
            e.stream, e.ws, e.ws16 = self.saved
            return self.ctx.__exit__(*a)

    def _predictor_join(self):
        """Main stream waits for everything issued on the predictor stream."""
        if self._pred_pending:
            torch.cuda.current_stream(self.device).wait_stream(self.pside)
            self._pred_pending = False

    # ------------------------------------------------------------------ thin launch helpers
    def _split(self, M, N, K):
        key = (M, N, K)
        s = self._split_cache.get(key)
        if s is None:
            s = (lib.vitae_gemm_pick_split_k if self.prec == PREC['fp32'] else lib.vitae_gemm_bf16x3_pick_split_k if self.prec == PREC['fp32x3']
                 else lib.vitae_gemm_bf16_pick_split_k)(M, N, K)
            while s > 1 and s * M * N > self.ws.numel():
                s -= 1
            self._split_cache[key] = s
        return s

    def _w16(self, w):
        """Device pointer of the bf16 shadow of arena weight ``w`` (0 when ``w`` is not an arena tensor)."""
        off = w.data_ptr() - self.params.data_ptr()
        if self.params16 is None or off < 0 or off >= self.n_total * 4:
            return 0
        return self.params16.data_ptr() + off // 2

    def _timed(self, flops, tag='other'):
        """bench.py's roofline instrumentation: HIP events around one GEMM launch; ``tag`` names the kernel family
        ('glds_pair' = gemm_glds_pair_kernel, 'glds' = gemm_glds_kernel, 'other' = gemm_bf16 / fp32 kernels)."""
        if self.gemm_timer is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.gemm_timer.append((a, b, flops, tag, self._scope))
        a.record()
        return b

    def _x3_ok(self, N, K, *ts, M=None) -> bool:
        """vitae_gemm_wsx3 serves this problem (every precondition of its launcher mirrored, so that a tiny batch or token count
        falls through to vitae_gemm_bf16x3 instead of aborting the step): 4-column groups everywhere, at least 8 rows
        and columns, 16-byte aligned arrays."""
        return (self.x3ws and N % 4 == 0 and K % 4 == 0 and N >= 8 and K >= 4 and (M is None or M >= 8)
                and all(t is None or t.data_ptr() % 16 == 0 for t in ts))

    def _x3_ws(self, side=False):
        """The wsx3 workspace of the stream the next launch goes to (main chain, predictor branch, wgrad side stream)."""
        if side:
            return self.ws_x3['wside']
        return self.ws_x3['pside' if self.stream == self.pside.cuda_stream else 'main']

    def _x3_split(self, M, N, K, ws):
        key = ('x3', M, N, K, ws.numel())
        s = self._split_cache.get(key)
        if s is None:
            s = lib.vitae_gemm_wsx3_pick_split_k(M, N, K)
            while s > 1 and lib.vitae_gemm_glds_ws_floats(M, N, s) > ws.numel():
                s -= 1
            self._split_cache[key] = s
        return s

    def _lin_fwd(self, x, w, bias, y, M, N, K, epi=EPI_NONE, aux=None, res=None):
        if self._x3_ok(N, K, x, w, y, bias, aux, res, M=M):
            ws = self._x3_ws()
            s = 1 if (epi & 15) == EPI_GELU else self._x3_split(M, N, K, ws)
            t = self._timed(2.0 * M * N * K)
            lib.vitae_gemm_wsx3(1, 1, _ptr(x), K, _ptr(w), K, _ptr(y), N, M, N, K, _ptr(bias), _ptr(res), N, epi, _ptr(aux), N, 0, s,
                                ws.data_ptr(), None, None, self.stream)
            if t is not None:
                t.record()
            return
        s = 1 if (epi & 15) == EPI_GELU else self._split(M, N, K)
        t = self._timed(2.0 * M * N * K)
        if self.prec == PREC['bf16']:
            w16 = self._w16(w)
            lib.vitae_gemm_bf16(1, 1, _ptr(x), K, w16 if w16 else _ptr(w), K, 1 if w16 else 0, _ptr(y), N, M, N, K,
                                _ptr(bias), _ptr(res), N, epi, _ptr(aux), N, 0, s, self.ws.data_ptr(), None, self.stream)
        else:
            lib.vitae_linear_fwd(self.prec, _ptr(x), _ptr(w), _ptr(bias), _ptr(y), M, N, K, epi, _ptr(aux), _ptr(res), s,
                                 self.ws.data_ptr(), self.stream)
        if t is not None:
            t.record()

    def _lin_bwd_x(self, dy, w, dx, M, N, K, epi=EPI_NONE, aux=None, accumulate=0, db=None):
        """dx = epi(dy @ W); with ``db`` the bias gradient colsum(dy) rides on the same launch (bf16 mode)
        or is a separate column-sum kernel (fp32 mode)."""
        if self._x3_ok(K, N, dy, w, dx, aux, M=M) and K >= 8:
            ws = self._x3_ws()
            s = self._x3_split(M, K, N, ws)
            t = self._timed(2.0 * M * N * K)
            lib.vitae_gemm_wsx3(1, 0, _ptr(dy), N, _ptr(w), K, _ptr(dx), K, M, K, N, None, None, 0, epi, _ptr(aux), K, accumulate, s,
                                ws.data_ptr(), None, None, self.stream)
            if db is not None:
                lib.vitae_colsum_accum(_ptr(dy), N, _ptr(db), M, N, self.stream)
            if t is not None:
                t.record()
            return
        s = self._split(M, K, N)
        t = self._timed(2.0 * M * N * K)
        if self.prec == PREC['bf16']:
            w16 = self._w16(w)
            lib.vitae_gemm_bf16(1, 0, _ptr(dy), N, w16 if w16 else _ptr(w), K, 1 if w16 else 0, _ptr(dx), K, M, K, N,
                                None, None, 0, epi, _ptr(aux), K, accumulate, s, self.ws.data_ptr(), _ptr(db), self.stream)
        else:
            lib.vitae_linear_bwd_input(self.prec, _ptr(dy), _ptr(w), _ptr(dx), M, N, K, epi, _ptr(aux), accumulate, s,
                                       self.ws.data_ptr(), self.stream)
            if db is not None:
                lib.vitae_colsum_accum(_ptr(dy), N, _ptr(db), M, N, self.stream)
        if t is not None:
            t.record()

    def _lin_bwd_w(self, dy, x, dw, db, M, N, K, tag=None):
        """dW (+)= dy^T x.  With ``tag`` (and overlap enabled) the GEMM is enqueued on the wgrad side stream:
        it only needs dy / x, which are final when this is called, and nobody reads dW before the end of the
        backward phase.  ``tag`` names the dy buffer so that ``_wg_fence(tag)`` can be placed in front of the
        kernel that next overwrites it."""
        s = self._split(N, K, M)
        side = tag is not None and self.overlap_wgrad and self.gemm_timer is None
        ws = self.ws
        if side and s > 1:      # a split reduction beside the chain needs scratch of its own (the chain's GEMMs use self.ws)
            ws = self.ws_wside
            while s > 1 and s * N * K > ws.numel():
                s -= 1
        if side:
            main = torch.cuda.current_stream(self.device)
            self.wside.wait_stream(main)
            stream = self.wside.cuda_stream
        else:
            stream = self.stream
        t = self._timed(2.0 * M * N * K)
        if self._x3_ok(K, M, dy, x, dw, db, M=N) and N % 4 == 0 and K >= 8:
            # dW = dy^T x (reduction over the M token rows: any multiple of 4), the bias gradient colsum(dy) by the same launch
            ws = self._x3_ws(side)       # (split-K tickets: never shared between streams)
            s3 = self._x3_split(N, K, M, ws)
            lib.vitae_gemm_wsx3(0, 0, _ptr(dy), N, _ptr(x), K, _ptr(dw), K, N, K, M, None, None, 0, EPI_NONE, None, 0, int(self._accum), s3,
                                ws.data_ptr(), None, _ptr(db), stream)
            db = None
            if self._epi_norm_on:        # this launch added the squares of dW to acc[GRADSQ] (vitae_gemm_glds_set_wgrad_sqnorm)
                off = (dw.data_ptr() - self.grads.data_ptr()) // 4
                if 0 <= off < self.tok_off:
                    # the kernel adds the squares of whatever it stores: a second launch into the same dW inside one grad window
                    # would count partial + total — every matrix has exactly one weight-gradient launch per step
                    if any(a < off + N * K and off < e for a, e in self._x3_cov):
                        raise VitaeError('fp32x3 gradient-norm coverage: two weight-gradient launches into one dW range in one step')
                    self._x3_cov.append((off, off + N * K))
        elif self.prec == PREC['bf16']:
            lib.vitae_gemm_bf16(0, 0, _ptr(dy), N, _ptr(x), K, 0, _ptr(dw), K, N, K, M, None, None, 0, EPI_NONE, None, 0,
                                int(self._accum), s, ws.data_ptr(), None, stream)
        else:
            lib.vitae_linear_bwd_weight(self.prec, _ptr(dy), _ptr(x), _ptr(dw), M, N, K, int(self._accum), s,
                                        ws.data_ptr(), stream)
        if t is not None:
            t.record()
        if db is not None:
            lib.vitae_colsum_accum(_ptr(dy), N, _ptr(db), M, N, stream)
        if side:
            ev = self._wg_events.get(tag)
            if ev is None:
                ev = self._wg_events[tag] = torch.cuda.Event()
            ev.record(self.wside)
            self._wg_pending.add(tag)

    def _lin_bwd(self, dy, w, x, dx, dw, db, M, N, K, epi=EPI_NONE, aux=None, dx_accumulate=0, tag=None):
        """Full backward of y = x W^T + b given dy: dx (+)= epi(dy W), dW (+)= dy^T x, db += colsum(dy).
        bf16 mode: ONE paired launch (dgrad + wgrad blocks side by side); fp32 mode: separate launches."""
        w16 = self._w16(w) if self.prec == PREC['bf16'] else 0
        if w16:
            t = self._timed(4.0 * M * N * K)   # dgrad + wgrad
            lib.vitae_linear_bwd_pair_bf16(_ptr(dy), w16, _ptr(x), _ptr(dx), _ptr(dw), _ptr(db), M, N, K, epi, _ptr(aux),
                                           dx_accumulate, int(self._accum), self.stream)
            if t is not None:
                t.record()
            return
        # the bias gradient colsum(dy) goes with the weight gradient (both only read dy; on the wgrad side stream when that overlaps):
        # as a separate launch on the main chain it was 84 x 11 us = 0.9 ms of the fp32-mode step
        w_db = db if ((tag is not None and self.overlap_wgrad) or self.x3ws) else None
        self._lin_bwd_w(dy, x, dw, w_db, M, N, K, tag=tag)
        self._lin_bwd_x(dy, w, dx, M, N, K, epi=epi, aux=aux, accumulate=dx_accumulate, db=None if w_db is not None else db)

    def _colsum_beside(self, x, ld, out, M, N, tag):
        """out[n] += colsum(x) on the wgrad side stream: a bias gradient that nothing on the main chain waits for until
        the end of the phase (``_wg_join``); ``tag`` names the buffer it reads, for ``_wg_fence``."""
        if not self.overlap_wgrad or self.gemm_timer is not None:
            lib.vitae_colsum_accum(_ptr(x), ld, _ptr(out), M, N, self.stream)
            return
        self.wside.wait_stream(torch.cuda.current_stream(self.device))
        lib.vitae_colsum_accum(_ptr(x), ld, _ptr(out), M, N, self.wside.cuda_stream)
        ev = self._wg_events.get(tag)
        if ev is None:
            ev = self._wg_events[tag] = torch.cuda.Event()
        ev.record(self.wside)
        self._wg_pending.add(tag)

    def _wg_fence(self, tag):
        """Make the current stream wait for the side-stream wgrad that still reads buffer ``tag``."""
        if tag in self._wg_pending:
            torch.cuda.current_stream(self.device).wait_event(self._wg_events[tag])
            self._wg_pending.discard(tag)

    def _wg_join(self):
        """All side-stream wgrads issued so far are complete for the current stream (end of a phase)."""
        if self._wg_pending:
            torch.cuda.current_stream(self.device).wait_stream(self.wside)
            self._wg_pending.clear()

    def _ln_fwd(self, x, pre, y, mean, rstd, M, D, y16=None):
        if _ABLATE_LN_FWD and pre.startswith(('blocks.', 'decoder_blocks.')):   # timing ablation only (results are garbage)
            return
        lib.vitae_layernorm_fwd(_ptr(x), _ptr(self.p[pre + 'weight']), _ptr(self.p[pre + 'bias']), _ptr(y), _ptr(y16),
                                _ptr(mean), _ptr(rstd), M, D, self.cfg.ln_eps, self.stream)

    def _ln_bwd(self, dy, x, pre, mean, rstd, dx, M, D, dx_accumulate, dx16=None, dx_colsum=None):
        """LayerNorm backward.  Default (round 4): no atomics — the launch leaves its d(gamma) / d(beta) / colsum(dx) partials as
        records in ``ln_part`` and ``_ln_flush`` (once per backward phase) adds the records of every LayerNorm of the phase into the
        gradient arena with one launch; nothing reads those gradients before the optimiser's tail."""
        # (few rows: the atomic kernel — its 3 D atomics per workgroup are cheaper than the reduce launches: batch 4 measured 10.8 us +
        # 4 x 13 us of reduces per step against 10.2 us)
        if self.ln_part_on and D in (256, 512, 768, 1024) and self.buf and M * D >= self.ln_part_min:
            G = lib.vitae_layernorm_bwd_part_records(M)
            part = self.buf.get('lnpart.' + pre)
            if part is None or part.numel() < G * 3 * D:      # (a LayerNorm the workspace does not know: allocated on first use)
                part = self.buf['lnpart.' + pre] = torch.empty(G * 3 * D, dtype=torch.float32, device=self.device)
            lib.vitae_layernorm_bwd_part(_ptr(dy), _ptr(x), _ptr(self.p[pre + 'weight']), _ptr(mean), _ptr(rstd), _ptr(dx),
                                         _ptr(part), _ptr(dx16), M, D, dx_accumulate, self.stream)
            self._ln_pending.append((part.data_ptr(), self.g[pre + 'weight'].data_ptr(), self.g[pre + 'bias'].data_ptr(),
                                     dx_colsum.data_ptr() if dx_colsum is not None else 0, G, D))
            return
        lib.vitae_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(self.p[pre + 'weight']), _ptr(mean), _ptr(rstd), _ptr(dx),
                                _ptr(self.g[pre + 'weight']), _ptr(self.g[pre + 'bias']), _ptr(dx16), _ptr(dx_colsum),
                                M, D, dx_accumulate, self.stream)

    def _ln_flush(self, final=False):
        """Add the pending LayerNorm partial records into the gradient arena: once per step, at the end of the last backward phase
        (``final``; nothing reads these vector gradients earlier — the optimiser's tail and the data-parallel exchange of the
        token / vector bucket both follow it), or at the end of every phase (``ln_flush_once`` off)."""
        pend = self._ln_pending
        if not pend or (self.ln_flush_once and not final):
            return
        u64 = lambda k: np.array([t[k] for t in pend], dtype=np.uint64)
        i32 = lambda k: np.array([t[k] for t in pend], dtype=np.int32)
        a_p, a_w, a_b, a_c, a_g, a_d = u64(0), u64(1), u64(2), u64(3), i32(4), i32(5)
        lib.vitae_ln_grad_reduce(len(pend), a_p.ctypes.data, a_w.ctypes.data, a_b.ctypes.data, a_c.ctypes.data, a_g.ctypes.data,
                                 a_d.ctypes.data, self.stream)
        self._ln_pending = []

    # ------------------------------------------------------------------ bf16-activation GEMM helpers (LDS-DMA kernel)
    def _g16_fwd(self, x16, w, bias, M, N, K, y=None, y16=None, epi=EPI_NONE, aux=None, res=None, name=None):
        """y / y16 = epi(x16 @ W16^T + b) (+ res) on the LDS-DMA GEMM (bf16 operands in HBM)."""
        key = ('g', M, N, K, self.ws16.numel())
        s = self._split_cache.get(key)
        if s is None:
            s = 1 if (epi & 15) == EPI_GELU else lib.vitae_gemm_glds_pick_split_k(M, N, K)
            while s > 1 and lib.vitae_gemm_glds_ws_floats(M, N, s) > self.ws16.numel():
                s -= 1
            self._split_cache[key] = s
        w2 = self._w2.get(name) if name is not None else None
        if w2 is not None:
            # two-plane weight ([W hi | W lo] side by side): the two-plane 64 x 64 workgroup, or a big tile over 2 K with x16 wrapping
            key2 = ('w2', M, N, K, self.ws16.numel())
            s2 = self._split_cache.get(key2)
            if s2 is None:
                s2 = 1 if (epi & 15) == EPI_GELU else lib.vitae_gemm_glds_w2_pick_split_k(M, N, K)
                while s2 > 1 and lib.vitae_gemm_glds_ws_floats(M, N, s2) > self.ws16.numel():
                    s2 -= 1
                self._split_cache[key2] = s2
            # (algorithmic FLOPs: the second plane's MFMAs are not counted as work; the tag follows the planner of the two-plane form:
            # the 64 x 64 two-plane workgroup, or a big tile over 2 K)
            t = self._timed(2.0 * M * N * K, self._w2_tag(M, N, K))
            lib.vitae_gemm_glds_w2(_ptr(x16), K, _ptr(w2), _ptr(y), N, _ptr(y16), N, M, N, K, _ptr(bias), _ptr(res), N,
                                   epi, _ptr(aux), N, 0, s2, self.ws16.data_ptr(), None, self.stream)
            if t is not None:
                t.record()
            return
        t = self._timed(2.0 * M * N * K, self._gemm_tag(1, 1, M, N, K, s, 'glds' if N < 8192 else 'glds_wide'))   # wide = the 64x128-tile instantiation
        lib.vitae_gemm_glds(1, 1, _ptr(x16), K, self._w16(w), K, _ptr(y), N, _ptr(y16), N, M, N, K, _ptr(bias), _ptr(res), N,
                            epi, _ptr(aux), N, 0, s, self.ws16.data_ptr(), None, self.stream)
        if t is not None:
            t.record()

    def _w2_tag(self, M, N, K):
        if self.gemm_timer is None:
            return 'ws64'
        key = ('w2tag', M, N, K)
        tag = self._split_cache.get(key)
        if tag is None:
            c = lib.vitae_gemm_glds_bt_choice(1, 1, M, N, 2 * K)       # (a big tile serves the planes as ONE reduction over 2 K)
            tag = self._split_cache[key] = {0: 'bt256', 3: 'bt128', 4: 'ws128'}.get(c, 'ws64')
        return tag

    def _gemm_tag(self, akc, bkc, M, N, K, split, default):
        """Instrumentation only: which kernel family serves this problem (csrc/gemm_glds.hip's planner)."""
        if self.gemm_timer is None:
            return default
        key = ('tag', akc, bkc, M, N, K, split)
        tag = self._split_cache.get(key)
        if tag is None:
            c = lib.vitae_gemm_glds_bt_choice(akc, bkc, M, N, K)
            tag = default if (c < 0 or lib.vitae_gemm_glds_pick_split_k(M, N, K) != split) else {0: 'bt256', 3: 'bt128', 4: 'ws128', 5: 'ws64'}.get(c, 'bt128')
            self._split_cache[key] = tag
        return tag

    def _wire_of(self, gview: torch.Tensor):
        """bf16 data-parallel exchange: address of this gradient tensor's slot in the wire buffer (same layout as the
        arena), so the wgrad epilogue writes the rounded copy itself and the bucket needs no separate rounding pass."""
        w = self.grads_wire16
        if w is None:
            return None
        return w.data_ptr() + (gview.data_ptr() - self.grads.data_ptr()) // 2

    def wire_uncovered_ranges(self, extra_covered=()):
        """Element ranges of the arena whose bf16 wire copy is NOT written by a GEMM epilogue (the reducer rounds those).
        ``extra_covered``: names to treat as covered as well (``_epi_norm_uncovered``: the predictor's matrices get their squares
        from the weight-gradient epilogues but no wire copy)."""
        if not self.act16:
            return [(0, self.n_total)]
        cfg = self.cfg
        covered = ['patch_embed.proj.weight', 'decoder_pred.weight', 'decoder_embed.weight'] + list(extra_covered)
        for pre, depth in (('blocks.', cfg.depth), ('decoder_blocks.', cfg.decoder_depth)):
            for i in range(depth):
                covered += [f'{pre}{i}.{n}.weight' for n in ('attn.qkv', 'attn.proj', 'mlp.fc1', 'mlp.fc2')]
        cov = set(covered)
        out, cur = [], None
        for n, (o, shp) in self.layout.items():
            k = (int(np.prod(shp)) + 3) // 4 * 4
            if n in cov:
                if cur is not None:
                    out.append(tuple(cur)); cur = None
            else:
                if cur is None:
                    cur = [o, o + k]
                else:
                    cur[1] = o + k
        if cur is not None:
            out.append(tuple(cur))
        return out

    def _g16_bwd(self, dy16, w, x16, dw, M, Mpad, N, K, dx=None, dx16=None, epi=EPI_NONE, aux=None, dx_colsum=None,
                 dy_colsum=None, dx_accumulate=0):
        """dx / dx16 = epi(dy16 @ W16), dW (+)= dy16^T @ x16 in one paired launch.  ``dw`` None: the input gradient only (the
        weight gradient is collected by ``_wgrad_group``)."""
        key = ('p', M, N, K, self.ws16.numel())
        s = self._split_cache.get(key)
        if s is None:
            s = lib.vitae_linear_bwd_pair_pick_split_k(M, Mpad, N, K)
            while s > 1 and lib.vitae_gemm_glds_ws_floats(M, K, s) > self.ws16.numel():
                s -= 1
            self._split_cache[key] = s
        tag = 'glds_pair' if N < 8192 else 'glds_pair_wide'
        if self.gemm_timer is not None:
            cd, cw = lib.vitae_gemm_glds_bt_choice(1, 0, M, K, N), lib.vitae_gemm_glds_bt_choice(0, 0, N, K, Mpad)
            if cd == 5 and cw == 5:
                tag = 'ws64_pair'   # both halves as wave-specialised 64 x 64 workgroups of one launch
            elif cd in (0, 3, 4) or cw in (0, 3, 4):
                tag = 'bt_bwd'      # the halves leave as two launches, at least one of them on a big tile
        if dw is None:
            tag = self._gemm_tag(1, 0, M, K, N, lib.vitae_gemm_glds_pick_split_k(M, K, N), 'glds')
        t = self._timed((4.0 if dw is not None else 2.0) * M * N * K, tag)
        lib.vitae_linear_bwd_pair_glds(_ptr(dy16), self._w16(w), _ptr(x16), _ptr(dx), _ptr(dx16), _ptr(dw),
                                       None if dw is None else self._wire_of(dw), M, Mpad,
                                       N, K,
                                       epi, _ptr(aux), _ptr(dx_colsum), _ptr(dy_colsum), int(dx_accumulate), int(self._accum), s,
                                       self.ws16.data_ptr(), self.ws16.numel(),
                                       self.stream)
        if t is not None:
            t.record()

    def _wgrad_group(self, items, M, Mpad, bias=None):
        """The weight gradients of one block's Linears in ONE launch (csrc/gemm_bt.hip: gemm_bt_wgrad_group_kernel).
        ``items``: [(dy16, x16, dw, N, K)]; ``bias``: {index: bias-gradient tensor} for column sums of dy taken beside it."""
        n = len(items)
        arr = lambda v: np.array(v, dtype=np.uint64)
        a_dy, a_x = arr([it[0].data_ptr() for it in items]), arr([it[1].data_ptr() for it in items])
        a_dw = arr([it[2].data_ptr() for it in items])
        wires = [self._wire_of(it[2]) for it in items]
        a_16 = arr([w or 0 for w in wires]) if any(wires) else None
        a_db = arr([(bias[i].data_ptr() if bias and i in bias else 0) for i in range(n)]) if bias else None
        Ns, Ks = np.array([it[3] for it in items], dtype=np.int32), np.array([it[4] for it in items], dtype=np.int32)
        t = self._timed(sum(2.0 * Mpad * it[3] * it[4] for it in items), 'bt_group')
        lib.vitae_wgrad_group_bt(n, a_dy.ctypes.data, a_x.ctypes.data, a_dw.ctypes.data, None if a_16 is None else a_16.ctypes.data,
                                 None if a_db is None else a_db.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, M, Mpad, int(self._accum),
                                 self.ws16.data_ptr(), self.ws16.numel(), self.stream)
        if t is not None:
            t.record()

    # ------------------------------------------------------------------ transformer block
    def _block_fwd16(self, pre, q, x_in, x_out, Bs, N, d, heads, hd, hid):
        """model/vit.py:139-144 with bf16 GEMM operands written by their producers."""
        b, p, M = self.buf, self.p, Bs * N
        q16 = self._qkv16_ok(N, hd)
        qkv32, qkv16 = (None, b[q + 'qkv_16']) if q16 else (b[q + 'qkv'], None)
        self._ln_fwd(x_in, pre + 'norm1.', None, b[q + 'mean1'], b[q + 'rstd1'], M, d, y16=b[q + 'y1_16'])
        self._g16_fwd(b[q + 'y1_16'], p[pre + 'attn.qkv.weight'], p[pre + 'attn.qkv.bias'], M, 3 * d, d, y=qkv32, y16=qkv16,
                      name=pre + 'attn.qkv.weight')
        t = self._timed(4.0 * Bs * heads * N * N * hd, 'attn')
        if q16:
            lib.vitae_sdpa_mfma_fwd_bf16in(_ptr(qkv16), _ptr(b[q + 'o']), _ptr(b[q + 'o_16']), _ptr(b[q + 'lse']), Bs, N, heads, hd,
                                           self.stream)
        else:
            lib.vitae_sdpa_mfma_fwd(_ptr(b[q + 'qkv']), _ptr(b[q + 'o']), _ptr(b[q + 'o_16']), _ptr(b[q + 'lse']), Bs, N, heads, hd,
                                    self.stream)
        if t is not None:
            t.record()
        self._g16_fwd(b[q + 'o_16'], p[pre + 'attn.proj.weight'], p[pre + 'attn.proj.bias'], M, d, d, y=b[q + 'xmid'], res=x_in,
                      name=pre + 'attn.proj.weight')
        self._ln_fwd(b[q + 'xmid'], pre + 'norm2.', None, b[q + 'mean2'], b[q + 'rstd2'], M, d, y16=b[q + 'y2_16'])
        self._g16_fwd(b[q + 'y2_16'], p[pre + 'mlp.fc1.weight'], p[pre + 'mlp.fc1.bias'], M, hid, d, y16=b[q + 'act_16'],
                      epi=EPI_GELU | self._aux16, aux=b[q + 'hpre'], name=pre + 'mlp.fc1.weight')
        self._g16_fwd(b[q + 'act_16'], p[pre + 'mlp.fc2.weight'], p[pre + 'mlp.fc2.bias'], M, d, hid, y=x_out, res=b[q + 'xmid'],
                      name=pre + 'mlp.fc2.weight')

    def _qkv16_ok(self, N, hd) -> bool:
        """bf16-only qkv for this stack: the flag and an MFMA head size (round 3: any sequence length — the two streaming backward
        kernels read the bf16 copy as well; VITAE_QKV_BF16_LONG=0 keeps fp32 q | k | v when the head does not fit LDS)"""
        if not (self.qkv16 and hd in (32, 64)):
            return False
        return bool(lib.vitae_sdpa_bwd_fused_fits(N, hd) or os.environ.get('VITAE_QKV_BF16_LONG', '1') != '0')

    def _dqkv32(self, dqkv, N, hd):
        """fp32 dqkv is write-only on the bf16-operand path (the qkv weight / input gradients read the bf16 copy): skip it
        whenever the one-launch attention backward applies (its LDS budget: whole head resident)."""
        NP = (N + 31) // 32 * 32
        lds = 4 * NP * (hd + 8) * 2 + 2 * NP * 4 + 3 * hd * 4
        return None if (lds <= 150 * 1024 and os.environ.get('VITAE_ATTN_BWD_FUSED', '1') != '0') else dqkv

    def _dyset(self, i, Mp, d) -> str:
        """Suffix of the buffer set that holds block i's output-gradient operands (side-stream grouped weight gradients: two
        alternating sets, so that block i's weight-gradient launch may still read its set while block i - 1 runs)."""
        return '_alt' if (self.wgrad_group_side and (i & 1) and self._grouped(Mp, d)) else ''

    def _block_bwd16(self, pre, q, s, x_in, Bs, N, d, heads, hd, hid, Mp, prev_fc2_bias, idx=0):
        """Backward of one block on bf16 operands.  On entry buf[s+'dx'] (fp32) and buf[s+'dx_16'] hold the
        output gradient and the fc2 bias gradient has already been produced by whoever wrote dx.
        ``prev_fc2_bias``: gradient slot of the fc2 bias of the block this one feeds INTO dx for (block i-1)."""
        b, p, g, M = self.buf, self.p, self.g, Bs * N
        self._scope = s
        sfx, sfx_out = self._dyset(idx, Mp, d), self._dyset(idx - 1, Mp, d)
        dx, dx16, dh16, dy, do, dqkv, dqkv16 = (b[s + 'dx'], b[s + 'dx_16' + sfx], b[s + 'dh_16' + sfx], b[s + 'dy'], b[s + 'do'],
                                                  b[s + 'dqkv'], b[s + 'dqkv_16' + sfx])
        dx16_out = b[s + 'dx_16' + sfx_out]          # norm1's bf16 result: the output gradient of block idx - 1
        # Many token rows (batch >= ~16, patch 8): the four weight gradients of the block leave as ONE launch at its end
        # (``_wgrad_group``) and the Linears' backward launches compute the input gradients only.  Each dy operand then has to
        # survive until that launch: the gradient w.r.t. the block's middle (norm2's output) goes to a second bf16 buffer.
        grp = self._grouped(Mp, d)
        side = grp and self.wgrad_group_side and self.gemm_timer is None
        dw = (lambda name: None) if grp else (lambda name: g[name])
        dmid16 = b[s + 'dx_16b' + sfx] if grp else dx16
        if side:
            self._wg_fence(f'{s}grp{idx & 1}')      # the weight gradients of block idx + 2 read this set
        # fc1's bias gradient colsum(dh): paired launches (few token rows) take it as row sums of dh16^T in the fc1 weight-gradient
        # workgroups (one more MFMA against ones in the first column tile, ONE atomic per output) — as column sums of the fc2
        # input-gradient epilogue it was 128 atomics per 64 x 64 tile: +1.6 / +3.9 us on the 8.7 / 7.4 us launches of the batch-4
        # step (tools/epi_ablate.py, round 5).  Grouped weight gradients (many rows): the big-tile epilogue folds a workgroup's
        # column sums in LDS first.
        db_w = not grp and self.fc1_bias_by_wgrad
        self._g16_bwd(dx16, p[pre + 'mlp.fc2.weight'], b[q + 'act_16'], dw(pre + 'mlp.fc2.weight'), M, Mp, d, hid,
                      dx16=dh16, epi=EPI_DGELU | self._aux16, aux=b[q + 'hpre'], dx_colsum=None if db_w else g[pre + 'mlp.fc1.bias'])
        self._g16_bwd(dh16, p[pre + 'mlp.fc1.weight'], b[q + 'y2_16'], dw(pre + 'mlp.fc1.weight'), M, Mp, hid, d, dx=dy,
                      dy_colsum=g[pre + 'mlp.fc1.bias'] if db_w else None)
        self._ln_bwd(dy, b[q + 'xmid'], pre + 'norm2.', b[q + 'mean2'], b[q + 'rstd2'], dx, M, d, 1, dx16=dmid16,
                     dx_colsum=g[pre + 'attn.proj.bias'])
        self._g16_bwd(dmid16, p[pre + 'attn.proj.weight'], b[q + 'o_16'], dw(pre + 'attn.proj.weight'), M, Mp, d, d, dx=do)
        t = self._timed(10.0 * Bs * heads * N * N * hd, 'attn')
        # grouped weight gradients: the qkv bias gradient colsum(dqkv) is collected by the attention backward itself (wave
        # shuffles + one atomic per column and workgroup) — as a separate bf16 column-sum launch it was 9-10 us on the main chain of
        # each of the 20 blocks at batch 32 / patch 8
        qkv_db = g[pre + 'attn.qkv.bias'] if (grp and self.attn_bias_colsum) else None
        if self._qkv16_ok(N, hd):
            lib.vitae_sdpa_mfma_bwd_bf16in(_ptr(b[q + 'qkv_16']), _ptr(b[q + 'o']), _ptr(do), _ptr(b[q + 'lse']), None, _ptr(dqkv16),
                                           _ptr(qkv_db), _ptr(b['delta']),
                                           Bs, N, heads, hd, self.stream)
        else:
            lib.vitae_sdpa_mfma_bwd(_ptr(b[q + 'qkv']), _ptr(b[q + 'o']), _ptr(do), _ptr(b[q + 'lse']), _ptr(self._dqkv32(dqkv, N, hd)),
                                    _ptr(dqkv16), None, _ptr(b['delta']), Bs, N, heads, hd, self.stream)
        if t is not None:
            t.record()
        # the qkv bias gradient colsum(dqkv) rides on the wgrad workgroups (one extra MFMA against a ones operand)
        self._g16_bwd(dqkv16, p[pre + 'attn.qkv.weight'], b[q + 'y1_16'], dw(pre + 'attn.qkv.weight'), M, Mp, 3 * d, d, dx=dy,
                      dy_colsum=None if grp else g[pre + 'attn.qkv.bias'])
        if grp:       # (before norm1's backward overwrites dx16, the fc2 weight gradient's dy operand)
            items = [(dx16, b[q + 'act_16'], g[pre + 'mlp.fc2.weight'], d, hid),
                     (dh16, b[q + 'y2_16'], g[pre + 'mlp.fc1.weight'], hid, d),
                     (dmid16, b[q + 'o_16'], g[pre + 'attn.proj.weight'], d, d),
                     (dqkv16, b[q + 'y1_16'], g[pre + 'attn.qkv.weight'], 3 * d, d)]
            bias = None if (qkv_db is not None and self._qkv16_ok(N, hd)) else {3: g[pre + 'attn.qkv.bias']}
            if side:
                # beside the chain: its launches leave a third of the CUs idle, and nothing reads these gradients before the phase ends
                tag = f'{s}grp{idx & 1}'
                self.wside.wait_stream(torch.cuda.current_stream(self.device))
                saved = (self.stream, self.ws16)
                self.stream, self.ws16 = self.wside.cuda_stream, self.ws16_wside
                try:
                    with torch.cuda.stream(self.wside):
                        self._wgrad_group(items, M, Mp, bias=bias)
                finally:
                    self.stream, self.ws16 = saved
                ev = self._wg_events.get(tag)
                if ev is None:
                    ev = self._wg_events[tag] = torch.cuda.Event()
                ev.record(self.wside)
                self._wg_pending.add(tag)
                self._wg_fence(f'{s}grp{(idx + 1) & 1}')    # norm1 writes the OTHER set's dx16, which block idx + 1's launch reads
            else:
                self._wgrad_group(items, M, Mp, bias=bias)
        self._ln_bwd(dy, x_in, pre + 'norm1.', b[q + 'mean1'], b[q + 'rstd1'], dx, M, d, 1, dx16=dx16_out,
                     dx_colsum=prev_fc2_bias)
        self._scope = None

    def _block_fwd(self, pre, q, x_in, x_out, Bs, N, d, heads, hd, hid):
        """model/vit.py:139-144.  pre = state-dict prefix, q = workspace prefix."""
        if self.act16:
            self._scope = q[:3]
            try:
                return self._block_fwd16(pre, q, x_in, x_out, Bs, N, d, heads, hd, hid)
            finally:
                self._scope = None
        b, p, M = self.buf, self.p, Bs * N
        self._ln_fwd(x_in, pre + 'norm1.', b[q + 'y1'], b[q + 'mean1'], b[q + 'rstd1'], M, d)
        self._lin_fwd(b[q + 'y1'], p[pre + 'attn.qkv.weight'], p[pre + 'attn.qkv.bias'], b[q + 'qkv'], M, 3 * d, d)
        if self.prec == PREC['bf16'] and hd in (32, 64):
            lib.vitae_sdpa_mfma_fwd(_ptr(b[q + 'qkv']), _ptr(b[q + 'o']), None, _ptr(b[q + 'lse']), Bs, N, heads, hd, self.stream)
        else:
            lib.vitae_sdpa_fwd(_ptr(b[q + 'qkv']), _ptr(b[q + 'o']), _ptr(b[q + 'lse']), Bs, N, heads, hd, self.stream)
        self._lin_fwd(b[q + 'o'], p[pre + 'attn.proj.weight'], p[pre + 'attn.proj.bias'], b[q + 'xmid'], M, d, d,
                      res=x_in)
        self._ln_fwd(b[q + 'xmid'], pre + 'norm2.', b[q + 'y2'], b[q + 'mean2'], b[q + 'rstd2'], M, d)
        self._lin_fwd(b[q + 'y2'], p[pre + 'mlp.fc1.weight'], p[pre + 'mlp.fc1.bias'], b[q + 'act'], M, hid, d,
                      epi=EPI_GELU | self._auxd, aux=b[q + 'hpre'])
        self._lin_fwd(b[q + 'act'], p[pre + 'mlp.fc2.weight'], p[pre + 'mlp.fc2.bias'], x_out, M, d, hid,
                      res=b[q + 'xmid'])

    def _block_bwd(self, pre, q, s, x_in, Bs, N, d, heads, hd, hid):
        """Backward of one block; the running gradient lives in buf[s+'dx'] and is updated in place."""
        b, p, g, M = self.buf, self.p, self.g, Bs * N
        dx, dh, dy, do, dqkv = b[s + 'dx'], b[s + 'dh'], b[s + 'dy'], b[s + 'do'], b[s + 'dqkv']
        # mlp.fc2 / fc1
        self._wg_fence(s + 'dh')                       # previous block's fc1 wgrad may still read dh
        self._lin_bwd(dx, p[pre + 'mlp.fc2.weight'], b[q + 'act'], dh, g[pre + 'mlp.fc2.weight'], g[pre + 'mlp.fc2.bias'],
                      M, d, hid, epi=EPI_DGELU | self._auxd, aux=b[q + 'hpre'], tag=s + 'dx')
        self._lin_bwd(dh, p[pre + 'mlp.fc1.weight'], b[q + 'y2'], dy, g[pre + 'mlp.fc1.weight'], g[pre + 'mlp.fc1.bias'],
                      M, hid, d, tag=s + 'dh')
        self._wg_fence(s + 'dx')                       # fc2 wgrad reads dx; LN backward updates it in place
        self._ln_bwd(dy, b[q + 'xmid'], pre + 'norm2.', b[q + 'mean2'], b[q + 'rstd2'], dx, M, d, 1)
        # attn.proj
        self._lin_bwd(dx, p[pre + 'attn.proj.weight'], b[q + 'o'], do, g[pre + 'attn.proj.weight'], g[pre + 'attn.proj.bias'],
                      M, d, d, tag=s + 'dx')
        self._wg_fence(s + 'dqkv')                     # previous block's qkv wgrad may still read dqkv
        if self.prec == PREC['bf16'] and hd in (32, 64):
            lib.vitae_sdpa_mfma_bwd(_ptr(b[q + 'qkv']), _ptr(b[q + 'o']), _ptr(do), _ptr(b[q + 'lse']), _ptr(dqkv), None, None,
                                    _ptr(b['delta']), Bs, N, heads, hd, self.stream)
        else:
            lib.vitae_sdpa_bwd(_ptr(b[q + 'qkv']), _ptr(b[q + 'o']), _ptr(do), _ptr(b[q + 'lse']), _ptr(dqkv),
                               _ptr(b['delta']), Bs, N, heads, hd, self.stream)
        # attn.qkv
        self._lin_bwd(dqkv, p[pre + 'attn.qkv.weight'], b[q + 'y1'], dy, g[pre + 'attn.qkv.weight'], g[pre + 'attn.qkv.bias'],
                      M, 3 * d, d, tag=s + 'dqkv')
        self._wg_fence(s + 'dx')                       # proj wgrad reads dx
        self._ln_bwd(dy, x_in, pre + 'norm1.', b[q + 'mean1'], b[q + 'rstd1'], dx, M, d, 1)

    # ------------------------------------------------------------------ forward
    def forward(self, view1: torch.Tensor, view2: Optional[torch.Tensor], noise: torch.Tensor, mask_ratio: float,
                training: bool = True, defer_predictor_join: bool = False, defer_finalize: bool = False,
                loss_with_grad: bool = False):
        """Everything up to the four loss scalars and (contrastive) p1/p2.  ``noise`` is [Be, L]
        (view-1 rows first), the torch.rand of vit_autoenc.py:139.  ``loss_with_grad`` (the fused step, whose backward follows
        at once and whose gradient multipliers are already in ``hp``): the loss chain leaves the gradient w.r.t. the
        prediction in the same pass (csrc/loss_fused.hip) and ``backward_dec`` skips its own loss kernel."""
        cfg = self.cfg
        B = view1.shape[0]
        self._alloc(B, mask_ratio)
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        st, b, p = self.stream, self.buf, self.p
        Be, keep, L, D, Dd, P = self.Be, self.keep, cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim
        Ne, Nd, Me, Md = self.Ne, self.Nd, self.Me, self.Md
        C, (Lz, Hy, Wx), ps = cfg.in_chans, cfg.volume_size, cfg.patch_size
        for v in (view1, view2):
            if v is not None and (v.dtype != torch.float32 or not v.is_contiguous() or v.device != self.device
                                  or tuple(v.shape) != (B, C, Lz, Hy, Wx)):
                raise VitaeError(f'volumes must be contiguous fp32 [B,{C},{Lz},{Hy},{Wx}] on {self.device}')
        if cfg.contrastive and view2 is None:
            raise VitaeError('contrastive model needs view2')
        if tuple(noise.shape) != (Be, L) or noise.dtype != torch.float32 or not noise.is_contiguous():
            raise VitaeError(f'noise must be contiguous fp32 [{Be},{L}]')
        self.view1 = view1
        self.refresh_shadow()
        if self._head_done:
            self._head_done = False          # vitae_step_prologue zeroed acc and fetched hp for this step
        else:
            self.flush_hparams()
            lib.vitae_memset_zero(self.acc.data_ptr(), self.acc.numel() * 8, st)
        # --- target branch of the edge loss (blur + Sobel of the input, vit_autoenc.py:221-223) depends on the
        # data only: it runs on a side stream underneath the encoder/decoder and is joined before the edge MSE
        main = torch.cuda.current_stream(self.device)

        def target_branch(after=None):
            if after is not None:
                self.side.wait_event(after)
            else:
                self.side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side):
                ss = self.side.cuda_stream
                if self.target_one_pass and lib.vitae_target_edge_supported(C, len(self.taps), Lz, Hy, Wx):
                    # blur + Sobel of the input in one launch: no blurred intermediates in HBM (csrc/loss_fused.hip)
                    lib.vitae_target_edge(_ptr(view1), _ptr(b['edge_t']), self._taps_c, len(self.taps), B, C, Lz, Hy, Wx, ss)
                else:
                    lib.vitae_gauss_blur_fwd(_ptr(view1), _ptr(b['blur_tmp']), _ptr(b['blurred']), self._taps_c, len(self.taps),
                                             B * C, Lz, Hy, Wx, ss)
                    lib.vitae_sobel_edge_fwd(_ptr(b['blurred']), _ptr(b['edge_t']), None, None, B, C, Lz, Hy, Wx, ss)

        # where the branch forks off the main chain: 'start' (beside masking / gather / patch embedding), 'embed' (after the
        # patch-embedding GEMM, beside the first encoder blocks), 'decoder' (beside the first decoder blocks)
        fork = self.target_fork
        # 'start1' (experiment, measured WORSE): the same fork point, but the branch is ENQUEUED after the main chain's next kernel.
        # In a captured graph the child node created first keeps the parent's hardware queue and the other one pays a ~7-12 us
        # cross-queue hop; created second, the main chain does stay on its queue (and the patch-embedding GEMM runs 49 instead of
        # 91 us without the blur beside it) — but the graph executor then parks the branch on a queue whose earlier entries wait
        # for the latent: blur + Sobel only start at 1.75 ms, the loss waits for them, the step goes 4.72 -> 4.96 ms
        fork_ev = None
        if fork == 'start':
            target_branch()
        elif fork == 'start1':
            fork_ev = torch.cuda.Event()
            fork_ev.record(main)
        # --- masking, kept-patch gather, patch embedding, sequence assembly
        a16 = self.act16
        pat, pat16 = (None, b['patches_16']) if a16 else (b['patches'], None)
        lib.vitae_random_masking(_ptr(noise), _ptr(b['ids_shuffle']), _ptr(b['ids_restore']), _ptr(b['mask']),
                                 _ptr(b['ids_restore64']), Be, L, keep, st)
        if fork_ev is not None:
            target_branch(after=fork_ev)
        if cfg.contrastive:
            lib.vitae_gather_patches_2views(_ptr(view1), _ptr(view2), _ptr(b['ids_shuffle']), _ptr(pat), _ptr(pat16), B, C, Lz, Hy, Wx,
                                            ps, keep, st)
        else:
            lib.vitae_gather_patches(_ptr(view1), _ptr(b['ids_shuffle']), _ptr(pat), _ptr(pat16), B, C, Lz, Hy, Wx, ps, keep, st)
        if a16:
            self._g16_fwd(pat16, p['patch_embed.proj.weight'], p['patch_embed.proj.bias'], Be * keep, D, P, y=b['tok'])
        else:
            self._lin_fwd(b['patches'], p['patch_embed.proj.weight'], p['patch_embed.proj.bias'], b['tok'], Be * keep, D, P)
        if fork == 'embed':
            target_branch()
        ex = b['encx']
        lib.vitae_encoder_assemble_fwd(_ptr(b['tok']), _ptr(p['cls_token']), _ptr(self.buffers['pos_embed']),
                                       _ptr(b['ids_shuffle']), _ptr(ex[0]), Be, L, keep, D, st)
        for i in range(cfg.depth):
            self._block_fwd(f'blocks.{i}.', f'enc{i}.', ex[i], ex[i + 1], Be, Ne, D, cfg.num_heads, self.hd, self.Hm)
        self._ln_fwd(ex[cfg.depth], 'norm.', b['latent'], b['lat_mean'], b['lat_rstd'], Me, D, y16=b.get('latent_16'))
        if cfg.contrastive and self.overlap_predictor:
            # the predictor branch only needs the latent: it runs on its own stream beside the decoder and the loss chain
            self.pside.wait_stream(torch.cuda.current_stream(self.device))
            with self._OnPredictorStream(self):
                self._predictor_fwd(training, None)
            self._pred_pending = True
        if fork == 'decoder':
            target_branch()
        # --- decoder (view 1 only)
        if a16:
            self._g16_fwd(b['latent_16'], p['decoder_embed.weight'], p['decoder_embed.bias'], B * Ne, Dd, D, y=b['e'], name='decoder_embed.weight')
        else:
            self._lin_fwd(b['latent'], p['decoder_embed.weight'], p['decoder_embed.bias'], b['e'], B * Ne, Dd, D)
        dx_ = b['decx']
        lib.vitae_decoder_assemble_fwd(_ptr(b['e']), _ptr(p['mask_token']), _ptr(self.buffers['decoder_pos_embed']),
                                       _ptr(b['ids_restore']), _ptr(dx_[0]), B, L, keep, Dd, st)
        for i in range(cfg.decoder_depth):
            self._block_fwd(f'decoder_blocks.{i}.', f'dec{i}.', dx_[i], dx_[i + 1], B, Nd, Dd, cfg.decoder_num_heads,
                            self.hdd, self.Hmd)
        if a16:
            self._ln_fwd(dx_[cfg.decoder_depth], 'decoder_norm.', None, b['dn_mean'], b['dn_rstd'], Md, Dd, y16=b['dn_16'])
            self._g16_fwd(b['dn_16'], p['decoder_pred.weight'], p['decoder_pred.bias'], Md, P, Dd, y=b['predfull'], name='decoder_pred.weight')
        else:
            self._ln_fwd(dx_[cfg.decoder_depth], 'decoder_norm.', b['dn'], b['dn_mean'], b['dn_rstd'], Md, Dd)
            self._lin_fwd(b['dn'], p['decoder_pred.weight'], p['decoder_pred.bias'], b['predfull'], Md, P, Dd)
        # --- loss chain on pred = predfull[:, 1:, :]
        pred_ptr, pbs = b['predfull'].data_ptr() + P * 4, Nd * P
        torch.cuda.current_stream(self.device).wait_stream(self.side)   # edge map of the blurred target is ready
        self._loss_grad_done = bool(loss_with_grad and self.loss_one_pass and lib.vitae_loss_fwd_bwd_supported(C, Lz, Hy, Wx, ps))
        if self._loss_grad_done:
            # (bf16 step: the gradient leaves in bf16 only — decoder_pred's dgrad / wgrad read that copy, its bias gradient is the
            # column sum of the same copy)
            lib.vitae_loss_fwd_bwd(pred_ptr, pbs, _ptr(view1), _ptr(b['mask']), _ptr(b['edge_t']), _ptr(self.hp),
                                   None if a16 else b['dpredfull'].data_ptr() + P * 4, (b['dpred_16'].data_ptr() + P * 2) if a16 else None, None,
                                   _ptr(self.acc), self.mask_sum, B, C, Lz, Hy, Wx, ps, st)
        else:
            lib.vitae_loss_fwd_fused(pred_ptr, pbs, _ptr(view1), _ptr(b['mask']), _ptr(b['edge_t']), _ptr(b['pred_vol']),
                                     _ptr(b['edge_p']), _ptr(self.acc), B, C, Lz, Hy, Wx, ps, st)
        if not defer_finalize:
            self.loss_finalize()
        if cfg.contrastive and not self._pred_pending:
            self._predictor_fwd(training, st)            # not overlapped: in program order on the main stream
        if not defer_predictor_join:
            self._predictor_join()

    def loss_finalize(self):
        """acc -> losses[0:4] (vit_autoenc.py:231-232); nothing on the step's dependent chain reads it"""
        lib.vitae_loss_finalize(_ptr(self.acc), _ptr(self.hp), _ptr(self.losses), self.mask_sum, self.edge_count,
                                torch.cuda.current_stream(self.device).cuda_stream)

    def _predictor_fwd(self, training: bool, st):
        """predictor on both views (vit_autoenc.py:280-284); launches on ``self.stream``."""
        cfg, b, p = self.cfg, self.buf, self.p
        R, D = self.R, cfg.embed_dim
        p16 = self.pred16 and training
        if p16:
            self._g16_fwd(b['latent_16'], p['predictor.0.weight'], None, 2 * R, D, D, y=b['ph'])
        else:
            self._lin_fwd(b['latent'], p['predictor.0.weight'], None, b['ph'], 2 * R, D, D)
        for v in range(2):
            o = v * R * D * 4
            if not training:     # model.eval(): nn.BatchNorm1d normalises with its running statistics (vit_autoenc.py:263-268)
                lib.vitae_bn1d_relu_eval(b['ph'].data_ptr() + o, _ptr(p['predictor.1.weight']), _ptr(p['predictor.1.bias']),
                                         _ptr(self.buffers['predictor.1.running_mean']),
                                         _ptr(self.buffers['predictor.1.running_var']), b['pr'].data_ptr() + o, R, D, 1e-5,
                                         self.stream)
                continue
            args = (b['ph'].data_ptr() + o, _ptr(p['predictor.1.weight']), _ptr(p['predictor.1.bias']),
                    b['pr'].data_ptr() + o, (b['pr_16'].data_ptr() + o // 2) if p16 else None,
                    b['bn_mean'].data_ptr() + v * D * 4,
                    b['bn_rstd'].data_ptr() + v * D * 4,
                    _ptr(self.buffers['predictor.1.running_mean']) if training else None,
                    _ptr(self.buffers['predictor.1.running_var']) if training else None,
                    _ptr(self.buffers['predictor.1.num_batches_tracked']) if training else None,
                    R, D, 1e-5, 0.1)
            if 'bn_ws' in b:
                lib.vitae_bn1d_relu_fwd_split(*args, _ptr(b['bn_ws']), self.stream)
            else:
                lib.vitae_bn1d_relu_fwd(*args, self.stream)
        if p16:
            self._g16_fwd(b['pr_16'], p['predictor.3.weight'], p['predictor.3.bias'], 2 * R, D, D, y=b['pout'])
        else:
            self._lin_fwd(b['pr'], p['predictor.3.weight'], p['predictor.3.bias'], b['pout'], 2 * R, D, D)

    def contrastive_loss_fwd(self):
        """utils/train_one_epoch.py:113-114 on (p1, z2), (p2, z1); result -> losses[4]."""
        b, R, D = self.buf, self.R, self.cfg.embed_dim
        o = R * D * 4
        p1, p2, z1, z2 = b['pout'].data_ptr(), b['pout'].data_ptr() + o, b['latent'].data_ptr(), b['latent'].data_ptr() + o
        st = self.pside.cuda_stream if self._pred_pending else self.stream
        lib.vitae_cosine_loss_fwd(p1, z2, p2, z1, _ptr(self.acc), _ptr(self.hp), self.losses.data_ptr() + 16, R, D, st)

    def contrastive_loss_bwd(self):
        b, R, D = self.buf, self.R, self.cfg.embed_dim
        o = R * D * 4
        p1, p2, z1, z2 = b['pout'].data_ptr(), b['pout'].data_ptr() + o, b['latent'].data_ptr(), b['latent'].data_ptr() + o
        if self._pred_pending:
            # the branch's backward accumulates into gradient slots zeroed by begin_grad_window on the main stream
            self.pside.wait_stream(torch.cuda.current_stream(self.device))
            with self._OnPredictorStream(self):
                self._cosine_bwd(p1, z2, p2, z1, o)
                self._predictor_bwd()
            return
        self._cosine_bwd(p1, z2, p2, z1, o)

    # set by the fused step's cosine backward (which then writes bf16 dp_16 ONLY — buf['dp'] is dead on that route), consumed and cleared
    # by _predictor_bwd; any other producer of the predictor's output gradient fills fp32 buf['dp'] and leaves it False
    _dp16_ready = False

    def _cosine_bwd(self, p1, z2, p2, z1, o):
        b, R, D = self.buf, self.R, self.cfg.embed_dim
        if self.pred16:      # the predictor's backward reads bf16 only: the gradient leaves the loss kernel in that form
            lib.vitae_cosine_loss_bwd_bf16(p1, z2, p2, z1, _ptr(self.hp), None, None, b['dp_16'].data_ptr(), b['dp_16'].data_ptr() + o // 2,
                                           R, D, self.stream)
            self._dp16_ready = True
        else:
            lib.vitae_cosine_loss_bwd(p1, z2, p2, z1, _ptr(self.hp), b['dp'].data_ptr(), b['dp'].data_ptr() + o, R, D, self.stream)

    # ------------------------------------------------------------------ backward
    def begin_grad_window(self, accumulate: bool):
        """accumulate=False: first micro-step after zero_grad (matrix grads written with beta=0)."""
        self._accum = bool(accumulate)
        self._x3_cov = []
        if not accumulate:
            n = self.n_total - self.tok_off
            lib.vitae_memset_zero(self.grads.data_ptr() + self.tok_off * 4, n * 4,
                                  torch.cuda.current_stream(self.device).cuda_stream)

    def backward(self, have_dp: bool):
        """Reverse sweep.  hp[G_RECON], hp[G_EDGE] hold d total/d recon, d total/d raw_edge; when
        ``have_dp`` buf['dp'] holds d total / d [p1; p2].  1 + enc_chunks phases; after phase k the gradient
        range ``ddp.engine_bucket_ranges(self)[k]`` is final (bucket k may be all-reduced)."""
        self.backward_dec(have_dp)
        for hi, lo in self.enc_chunk_bounds():
            self.backward_enc(hi, lo)
        self.backward_tail()

    def backward_dec(self, have_dp: bool, part: Optional[str] = None):
        """loss chain, decoder, predictor, final encoder norm.  ``part``: None = all of it; 'top' = loss chain, decoder_pred,
        decoder_norm and the decoder blocks down to ``dec_cut`` (and the join of the predictor branch: a stream forked inside a
        captured phase must come back inside it); 'bottom' = the remaining blocks, sequence assembly, predictor / decoder_embed
        into the latent gradient, final encoder norm.  After 'top' the gradients of decoder_pred, the top blocks and the
        predictor are final (data-parallel bucket 0 of a two-bucket decoder)."""
        cfg = self.cfg
        top, bottom = part in (None, 'top'), part in (None, 'bottom')
        cut = self.dec_cut if part is not None else 0
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        st, b, p, g = self.stream, self.buf, self.p, self.g
        B, Be, keep, L, D, Dd, P = self.B, self.Be, self.keep, cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim
        Ne, Nd, Me, Md = self.Ne, self.Nd, self.Me, self.Md
        C, (Lz, Hy, Wx), ps = cfg.in_chans, cfg.volume_size, cfg.patch_size
        view1 = self.view1
        pred_ptr, dpred_ptr, pbs = b['predfull'].data_ptr() + P * 4, b['dpredfull'].data_ptr() + P * 4, Nd * P
        a16 = self.act16
        dx_ = b['decx']
        nd = cfg.decoder_depth
        blocks = [i for i in reversed(range(nd)) if (i >= cut and top) or (i < cut and bottom)]
        if top and self._loss_grad_done:
            self._loss_grad_done = False        # forward(loss_with_grad=True) left dpred / dpred_16 already
        elif top:
            lib.vitae_loss_bwd_fused(pred_ptr, _ptr(b['pred_vol']), _ptr(view1), _ptr(b['mask']), _ptr(b['edge_p']), _ptr(b['edge_t']),
                                     _ptr(self.hp), _ptr(b.get('dG')), dpred_ptr, (b['dpred_16'].data_ptr() + P * 2) if a16 else None,
                                     None, pbs, self.mask_sum, B, C, Lz, Hy, Wx, ps, st)
        if a16:
            if top:
                # decoder_pred (bias grad = column sum of the bf16 dpred, cls rows zero), decoder_norm -> dx, dx_16, fc2 bias grad of
                # the last block
                self._g16_bwd(b['dpred_16'], p['decoder_pred.weight'], b['dn_16'], g['decoder_pred.weight'], Md, self.Mpd, P, Dd,
                              dx=b['ddn'], dy_colsum=g['decoder_pred.bias'])
                self._ln_bwd(b['ddn'], dx_[nd], 'decoder_norm.', b['dn_mean'], b['dn_rstd'], b['decdx'], Md, Dd, 0,
                             dx16=b['decdx_16' + self._dyset(nd - 1, self.Mpd, Dd)],
                             dx_colsum=g[f'decoder_blocks.{nd - 1}.mlp.fc2.bias'])
            for i in blocks:
                self._block_bwd16(f'decoder_blocks.{i}.', f'dec{i}.', 'dec', dx_[i], B, Nd, Dd, cfg.decoder_num_heads, self.hdd,
                                  self.Hmd, self.Mpd, g[f'decoder_blocks.{i - 1}.mlp.fc2.bias'] if i > 0 else None, idx=i)
        else:
            if top:
                self._lin_bwd(b['dpredfull'], p['decoder_pred.weight'], b['dn'], b['ddn'], g['decoder_pred.weight'],
                              g['decoder_pred.bias'], Md, P, Dd)
                self._ln_bwd(b['ddn'], dx_[nd], 'decoder_norm.', b['dn_mean'], b['dn_rstd'], b['decdx'], Md, Dd, 0)
            for i in blocks:
                self._block_bwd(f'decoder_blocks.{i}.', f'dec{i}.', 'dec', dx_[i], B, Nd, Dd, cfg.decoder_num_heads, self.hdd,
                                self.Hmd)
        if part == 'top':
            if cfg.contrastive and have_dp and self._pred_pending:
                self._predictor_join()          # dph and the predictor's parameter gradients are final
                # ... and dph goes through predictor.0 HERE: this phase's bucket holds that weight, and its AdamW runs beside
                # the next phase
                self._pred0_dgrad()
                self._pred_joined = True
            self._ln_flush()
            self._wg_join()
            return
        lib.vitae_decoder_assemble_bwd(_ptr(b['decdx']), _ptr(b['ids_shuffle']), _ptr(b['de']), _ptr(b.get('de_16')),
                                       _ptr(g['mask_token']), B, L, keep, Dd, st)

        def dec_embed_bwd(accumulate):
            """decoder_embed: dW, db and dlatent[view-1 rows] (+)= de @ W."""
            if a16:     # dgrad + wgrad + bias gradient in one launch
                self._g16_bwd(b['de_16'], p['decoder_embed.weight'], b['latent_16'], g['decoder_embed.weight'], B * Ne, self.Mpl,
                              Dd, D, dx=b['dlatent'], dy_colsum=g['decoder_embed.bias'], dx_accumulate=accumulate)
            else:
                self._lin_bwd_w(b['de'], b['latent'], g['decoder_embed.weight'], None, B * Ne, Dd, D)
                self._lin_bwd_x(b['de'], p['decoder_embed.weight'], b['dlatent'], B * Ne, Dd, D, accumulate=accumulate,
                                db=g['decoder_embed.bias'])

        # predictor (both views) -> dlatent ; then decoder_embed adds into the view-1 rows
        if cfg.contrastive and have_dp:
            R = self.R
            if self._pred_joined:               # part 'top' joined the branch and took dph through predictor.0 already
                self._pred_joined = False
            else:
                if self._pred_pending:
                    self._predictor_join()      # dph and the predictor's parameter gradients are final
                else:
                    self._predictor_bwd()
                self._pred0_dgrad()
            dec_embed_bwd(1)
        else:
            if cfg.contrastive:   # predictor unused this step: its matrices get exact zeros
                for n in ('predictor.0.weight', 'predictor.3.weight'):
                    if not self._accum:
                        lib.vitae_memset_zero(g[n].data_ptr(), g[n].numel() * 4, st)
            if Be != B:
                lib.vitae_memset_zero(b['dlatent'].data_ptr(), b['dlatent'].numel() * 4, st)
            dec_embed_bwd(0)
        if a16:
            self._ln_bwd(b['dlatent'], b['encx'][cfg.depth], 'norm.', b['lat_mean'], b['lat_rstd'], b['encdx'], Me, D, 0,
                         dx16=b['encdx_16' + self._dyset(cfg.depth - 1, self.Mpe, D)],
                         dx_colsum=g[f'blocks.{cfg.depth - 1}.mlp.fc2.bias'])
        else:
            self._ln_bwd(b['dlatent'], b['encx'][cfg.depth], 'norm.', b['lat_mean'], b['lat_rstd'], b['encdx'], Me, D, 0)
        self._ln_flush()
        self._wg_join()

    def _pred0_dgrad(self):
        """dlatent = dph @ W0 (predictor.0 has no bias): the first writer of the latent gradient."""
        b, p, D = self.buf, self.p, self.cfg.embed_dim
        if self.pred16:
            self._g16_bwd(b['dph_16'], p['predictor.0.weight'], None, None, 2 * self.R, self.Mpe, D, D, dx=b['dlatent'])
        else:
            self._lin_bwd_x(b['dph'], p['predictor.0.weight'], b['dlatent'], 2 * self.R, D, D)

    def _predictor_bwd(self):
        """dp -> predictor.3, BatchNorm+ReLU, predictor.0 weight gradients and dph (launches on ``self.stream``)."""
        cfg, b, p, g = self.cfg, self.buf, self.p, self.g
        R, D = self.R, cfg.embed_dim
        p16 = self.pred16
        if p16:
            # predictor.3's dgrad + wgrad + bias gradient as one paired launch on bf16 operands.  dp_16: written by the fused step's cosine
            # backward; the autograd route hands the gradient over in fp32 (buf['dp'], model/vit_autoenc.py): cast it here
            if not self._dp16_ready:
                lib.vitae_cast_bf16(_ptr(b['dp']), _ptr(b['dp_16']), 2 * R * D, self.stream)
            self._dp16_ready = False
            self._g16_bwd(b['dp_16'], p['predictor.3.weight'], b['pr_16'], g['predictor.3.weight'], 2 * R, self.Mpe, D, D, dx=b['dpr'],
                          dy_colsum=g['predictor.3.bias'])
        else:
            self._lin_bwd_w(b['dp'], b['pr'], g['predictor.3.weight'], None, 2 * R, D, D)
            self._lin_bwd_x(b['dp'], p['predictor.3.weight'], b['dpr'], 2 * R, D, D, db=g['predictor.3.bias'])
        for v in range(2):
            o = v * R * D * 4
            args = (b['dpr'].data_ptr() + o, b['ph'].data_ptr() + o, b['pr'].data_ptr() + o,
                    _ptr(p['predictor.1.weight']), b['bn_mean'].data_ptr() + v * D * 4,
                    b['bn_rstd'].data_ptr() + v * D * 4, b['dph'].data_ptr() + o,
                    (b['dph_16'].data_ptr() + o // 2) if p16 else None,
                    _ptr(g['predictor.1.weight']), _ptr(g['predictor.1.bias']), R, D)
            if 'bn_ws' in b:
                lib.vitae_bn1d_relu_bwd_split(*args, _ptr(b['bn_ws']), self.stream)
            else:
                lib.vitae_bn1d_relu_bwd(*args, self.stream)
        if p16:      # dW0 = dph16^T @ latent16 (both row-contiguous bf16, reduced over the padded row count)
            lib.vitae_gemm_glds(0, 0, _ptr(b['dph_16']), D, _ptr(b['latent_16']), D, _ptr(g['predictor.0.weight']), D, None, D,
                                D, D, self.Mpe, None, None, 0, EPI_NONE, None, 0, int(self._accum), 1, None, None, self.stream)
        else:
            self._lin_bwd_w(b['dph'], b['latent'], g['predictor.0.weight'], None, 2 * R, D, D)

    def backward_enc(self, hi: int, lo: int):
        """encoder blocks hi, hi-1, ..., lo."""
        cfg = self.cfg
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        ex = self.buf['encx']
        for i in range(hi, lo - 1, -1):
            if self.act16:
                self._block_bwd16(f'blocks.{i}.', f'enc{i}.', 'enc', ex[i], self.Be, self.Ne, cfg.embed_dim, cfg.num_heads,
                                  self.hd, self.Hm, self.Mpe, self.g[f'blocks.{i - 1}.mlp.fc2.bias'] if i > 0 else None, idx=i)
            else:
                self._block_bwd(f'blocks.{i}.', f'enc{i}.', 'enc', ex[i], self.Be, self.Ne, cfg.embed_dim, cfg.num_heads,
                                self.hd, self.Hm)
        self._ln_flush()
        self._wg_join()

    def backward_tail(self):
        """sequence assembly and patch-embedding weight gradient (the input is data: no dgrad)."""
        cfg = self.cfg
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        b, g = self.buf, self.g
        T, D, P = self.Be * self.keep, cfg.embed_dim, cfg.patch_dim
        lib.vitae_encoder_assemble_bwd(_ptr(b['encdx']), _ptr(b['dtok']), _ptr(b['dtok_16']) if self.act16 else None,
                                       _ptr(g['cls_token']), self.Be, self.keep, D, self.stream)
        if self.act16:
            self._colsum_beside(b['dtok'], D, g['patch_embed.proj.bias'], T, D, 'dtok')     # beside the wgrad below
            # dW[D, P] = dtok16^T @ patches16 (both row-contiguous bf16, reduced over the padded token count)
            t = self._timed(2.0 * T * D * P, 'glds_wide')
            lib.vitae_gemm_glds(0, 0, _ptr(b['dtok_16']), D, _ptr(b['patches_16']), P, _ptr(g['patch_embed.proj.weight']), P,
                                self._wire_of(g['patch_embed.proj.weight']), P, D, P, self.Mpt, None, None, 0, EPI_NONE, None, 0,
                                int(self._accum), 1, None, None, self.stream)
            if t is not None:
                t.record()
        else:
            self._lin_bwd_w(b['dtok'], b['patches'], g['patch_embed.proj.weight'], g['patch_embed.proj.bias'], T, D, P)
        self._ln_flush(final=True)
        self._wg_join()

    # ------------------------------------------------------------------ optimiser
    # bf16 precision mode (round 6): AdamW's two moments are STORED in bf16 — computed in fp32 from the stored values, the unrounded
    # m_new / v_new enter the parameter update, only the write-back is rounded: 22 instead of 30 bytes of HBM traffic per parameter
    # (the optimiser was 0.65 ms of the 4.0 ms batch-4 step).  tools/opt_state_ablation.py: the reference's pinned ViT-B trajectory
    # moves by 2e-7..2e-6 and an 80-step loss curve by 1e-7 (another masking seed: 1e-3).  fp32 / fp32x3 modes keep fp32 moments;
    # VITAE_OPT_STATE16=0 keeps them in bf16 mode too.  Round-to-nearest would stall a moment whose update is under half an ulp, so
    # betas closer to 1 than 1 - 2^-6 (torch's default beta2 = 0.999) keep fp32 storage.
    _STATE16_MAX_BETA = 1.0 - 2.0 ** -6

    def _state16_ok(self, betas) -> bool:
        return (self.prec == PREC['bf16'] and os.environ.get('VITAE_OPT_STATE16', '1') != '0'
                and max(float(betas[0]), float(betas[1])) <= self._STATE16_MAX_BETA)

    def init_optimizer(self, weight_decay: float = 0.05, betas=(0.9, 0.95), eps: float = 1e-8):
        self.state16 = self._state16_ok(betas)
        dt = torch.bfloat16 if self.state16 else torch.float32
        self.opt_state = {'exp_avg': torch.zeros(self.n_total, dtype=dt, device=self.device),
                          'exp_avg_sq': torch.zeros(self.n_total, dtype=dt, device=self.device)}
        self.weight_decay, self.betas, self.eps = weight_decay, betas, eps
        self.write_opt_step(0)

    state16 = False

    def _adamw(self, o: int, n: int, gnorm, weight_decay: float, st, g16=None):
        """One AdamW launch over arena elements [o, o + n): fp32 or bf16 moments, fp32 gradients or the bf16 wire copy."""
        s = self.opt_state
        sh = (self.params16.data_ptr() + 2 * o) if self.params16 is not None else None
        es = s['exp_avg'].element_size()
        m, v = s['exp_avg'].data_ptr() + es * o, s['exp_avg_sq'].data_ptr() + es * o
        p = self.params.data_ptr() + 4 * o
        if self.state16:
            g = (g16.data_ptr() + 2 * o) if g16 is not None else (self.grads.data_ptr() + 4 * o)
            if gnorm is None:       # gated by the accumulator block itself (per-bucket launches: no finalisation node in front)
                lib.vitae_adamw_step_s16_acc(p, g, 1 if g16 is not None else 0, m, v, sh, n, _ptr(self.hp), _ptr(self.acc), weight_decay, st)
            else:
                lib.vitae_adamw_step_s16(p, g, 1 if g16 is not None else 0, m, v, sh, n, _ptr(self.hp), gnorm, weight_decay, st)
        elif g16 is not None:
            lib.vitae_adamw_step_bf16g(p, g16.data_ptr() + 2 * o, m, v, sh, n, _ptr(self.hp), gnorm, weight_decay, st)
        else:
            lib.vitae_adamw_step(p, self.grads.data_ptr() + 4 * o, m, v, sh, n, _ptr(self.hp), gnorm, weight_decay, st)

    def optimizer_hparams(self, lr: float):
        """Host side of one AdamW step: advances the step count, refreshes lr / bias corrections."""
        self.opt_step += 1          # the host's belief; the device count (hp[VITAE_HP_STEP], read_opt_step) is the truth
        b1, b2 = self.betas
        if self.state16 and max(float(b1), float(b2)) > self._STATE16_MAX_BETA:
            raise VitaeError(f'betas {self.betas}: the optimiser state was allocated in bf16 (betas <= {self._STATE16_MAX_BETA:.6f} at '
                             'init_optimizer); a moment this slow would stall under round-to-nearest — set VITAE_OPT_STATE16=0')
        # bc1, bc2 < 0: the AdamW kernels derive 1 - beta^t from the device-side count of APPLIED steps and from -(1 - beta), which
        # the host forms in double (csrc/optim.hip: device_bias_corrections)
        self.set_hparams(lr=lr, beta1=b1, beta2=b2, eps=self.eps, bc1=-(1.0 - float(b1)), bc2=-(1.0 - float(b2)))

    def grad_norm_and_step(self):
        """utils/misc.py:265-267: global grad L2 norm -> losses[5]; AdamW over the arena
        (decayed: matrices + tokens; not decayed: vectors)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.flush_hparams()
        gn = self.losses.data_ptr() + 20
        # bf16 data-parallel exchange: the reduced gradients live in the wire buffer (only on the fused-step route, which
        # raises the flag right after the exchange it issued)
        g16 = self.grads_wire16 if self._wire_ready else None
        if g16 is not None:
            lib.vitae_grad_sqnorm_bf16(g16.data_ptr(), self.n_total, _ptr(self.acc), gn, st)
        else:
            lib.vitae_grad_sqnorm(self.grads.data_ptr(), self.n_total, _ptr(self.acc), gn, st)
        self._adamw(0, self.vec_off, gn, self.weight_decay, st, g16)                              # matrices + tokens: decayed
        self._adamw(self.vec_off, self.n_total - self.vec_off, gn, 0.0, st, g16)                  # vectors: not
        lib.vitae_opt_count_bump(_ptr(self.hp), gn, st)
        self.refresh_w2(st)

    # --- optimiser inside the backward: the matrices of a gradient bucket are final when its backward phase ends
    # (data parallel: when its all-reduce has landed), so their share of the grad-norm pass and their AdamW update run on
    # a side stream underneath the remaining, latency-bound backward kernels instead of as 0.8 ms of HBM-bound work
    # after it.  The step-skip on a non-finite gradient (GradScaler.step) cannot wait for the global norm here: each
    # bucket keys its skip on the RUNNING norm of the buckets finished so far.  On this path a non-finite gradient
    # originates in the loss backward (the 0/0 of the Sobel magnitude at |g| = 0) and everything downstream of it — every
    # bucket — is then non-finite too, so the running norm decides exactly like the global one; only an overflow that
    # first appears in a LATER bucket would leave earlier buckets updated.
    def _optimizer_in_backward_ok(self) -> bool:
        """Single-process form: buckets are final when their phase ends."""
        return self.overlap_optimizer and self.opt_state is not None and not self._ddp_active

    _ddp_active = False      # the step runner exchanges gradient buckets between the phases
    _ddp_bucket_opt = False  # ... and issues _opt_bucket itself once a bucket's all-reduce has landed

    def _opt_bucket(self, k: int, wait_main: bool = True):
        """grad-norm share + AdamW of gradient bucket k (matrices only) on the optimiser stream."""
        from . import ddp
        s0, e0 = ddp.engine_bucket_ranges(self)[k]
        n = e0 - s0
        if n <= 0:
            return
        if wait_main:
            self.oside.wait_stream(torch.cuda.current_stream(self.device))
        st = self.oside.cuda_stream
        o = s0 * 4
        run = self.losses.data_ptr() + 24          # losses[6]: norm of the buckets finished so far
        g16 = self.grads_wire16 if self._wire_ready else None
        if g16 is not None:
            lib.vitae_grad_sqnorm_bf16(g16.data_ptr() + o // 2, n, _ptr(self.acc), run, st)
            self._adamw(s0, n, run, self.weight_decay, st, g16)
        else:
            if self._epi_norm_on:
                # the weight-gradient epilogues of this bucket already added their squares; only what no such epilogue writes
                # (the predictor's matrices) is still read here
                for a, e in self._epi_norm_uncovered():
                    a, e = max(a, s0), min(e, e0)
                    if e > a:
                        lib.vitae_grad_sqnorm(self.grads.data_ptr() + 4 * a, e - a, _ptr(self.acc), None, st)
                if self.state16 and self.adamw_acc_gate:
                    run = None       # the AdamW launch reads the accumulator itself
                else:
                    lib.vitae_grad_norm_finalize(_ptr(self.acc), run, st)
            else:
                lib.vitae_grad_sqnorm(self.grads.data_ptr() + o, n, _ptr(self.acc), run, st)
            self._adamw(s0, n, run, self.weight_decay, st)
        self.refresh_w2(st, s0, e0)          # lo planes of the two-plane weights this bucket holds (same stream, behind their update)
        self._opt_pending = True

    _epi_norm_on = False

    def _epi_norm_uncovered(self):
        """matrix ranges of the arena that no LDS-DMA weight-gradient epilogue writes (cached)"""
        if self.x3ws:
            # fp32x3: the matrix segment minus what this step's vitae_gemm_wsx3 weight-gradient launches covered so far (a bucket is
            # closed only after every weight gradient inside it has been enqueued)
            cov = sorted(self._x3_cov)
            out, pos = [], 0
            for a, e in cov:
                if a > pos:
                    out.append((pos, a))
                pos = max(pos, (e + 3) // 4 * 4)
            if pos < self.tok_off:
                out.append((pos, self.tok_off))
            return out
        u = getattr(self, '_epi_unc', None)
        if u is None:
            # predictor on the LDS-DMA GEMMs (pred16): predictor.3's weight gradient leaves through the paired launch and predictor.0's
            # through vitae_gemm_glds' dy^T x form — both add their squares to acc[GRADSQ] like every other wgrad epilogue, so reading
            # them here again would count them twice (tests/test_gpu_model.py::test_grad_norm_fused_equals_generic_with_contrastive_weight_one)
            extra = ('predictor.0.weight', 'predictor.3.weight') if (self.pred16 and self.cfg.contrastive) else ()
            u = self._epi_unc = [(a, min(e, self.tok_off)) for a, e in self.wire_uncovered_ranges(extra) if a < self.tok_off]
        return u

    def _opt_tail(self):
        """Tokens + vectors (whose gradients are accumulated atomically all through the backward) and the final norm."""
        main = torch.cuda.current_stream(self.device)
        if not self._ddp_bucket_opt:         # data parallel: the runner joins the optimiser stream between the graphs
            main.wait_stream(self.oside)
        self._opt_pending = False
        st = main.cuda_stream
        gn = self.losses.data_ptr() + 20
        s = self.opt_state
        sh = self.params16.data_ptr() if self.params16 is not None else 0
        ot = self.tok_off * 4
        g16 = self.grads_wire16 if self._wire_ready else None
        # norm share of tokens + vectors, the global norm (last workgroup), AdamW over both segments, the step count: 2 launches
        es = s['exp_avg'].element_size()
        lib.vitae_opt_tail(self.params.data_ptr() + ot, (g16.data_ptr() + ot // 2) if g16 is not None else self.grads.data_ptr() + ot,
                           1 if g16 is not None else 0, s['exp_avg'].data_ptr() + es * self.tok_off, s['exp_avg_sq'].data_ptr() + es * self.tok_off,
                           1 if self.state16 else 0,
                           (sh + ot // 2) if sh else None, self.vec_off - self.tok_off, self.n_total - self.vec_off, _ptr(self.hp),
                           _ptr(self.acc), gn, self.weight_decay, st)

    # ------------------------------------------------------------------ fused training step
    # encoder backward is cut into this many phases (= gradient buckets = optimiser-in-backward units)
    # (3: since AdamW got faster a third, smaller last bucket shortens the exposed tail: 5.15 -> 5.00 ms on one box, even on another)
    enc_chunks = int(os.environ.get('VITAE_ENC_CHUNKS', '3'))
    # loss forward sums + gradient in one pass over the prediction inside the fused step (csrc/loss_fused.hip; 4-channel volumes)
    loss_one_pass = os.environ.get('VITAE_LOSS_ONE_PASS', '1') != '0'
    _loss_grad_done = False
    # (padded token rows x model width) from which a block's four weight gradients leave as ONE grouped launch.  Measured
    # (volumes/s, grouped vs paired): batch 32 2766 vs 2480, batch 16 1906 vs 1802 (encoder 1792 x 768 grouped: 1906 vs 1843 without),
    # patch 8 328 vs 294; batch 8 1336 vs 1361-1372 when its decoder (1792 x 512) is grouped, 1314-1343 when everything is
    # Round 4 (side-stream grouped launch, wave-specialised input-gradient launches beside it): batch 8 5.40 -> 5.28 ms, batch 12 6.84 -> 6.51
    # with both stacks grouped; batch 4 neutral (4.14 vs 4.17): threshold between them.
    def _branch(self, name):
        """The stream of a branch of the step — or the CURRENT stream when the step is small (round 6).  A captured step whose
        branches fork is replayed by the HIP runtime on several hardware queues, and every edge between two queues is a signal
        the command processors hand over: with DEBUG_HIP_FORCE_GRAPH_QUEUES=1 (everything on one queue) the batch-4 step ran
        3.72 instead of 3.93 ms, config 5 3.78 / 4.02, ViT-L/16 128^3 8.86 / 9.18, batch 8 even — and batch 16 / 32 / patch 8
        lost 1.2 / 4.1 / 2.4 % (profiles/round6_graph_queues.txt): what runs beside the chain there is worth more than the
        hand-overs cost.  So the branches fork only from ``side_min_rows`` decoder token rows up (VITAE_SIDE_MIN_ROWS; VITAE_SIDE_STREAMS
        = all | none | a comma list forces the set); data parallel always forks (the runner joins the optimiser stream between
        the graphs of a step)."""
        if self.side_streams is None:
            on = self._ddp_active or getattr(self, 'Md', 0) >= self.side_min_rows
        else:
            on = name in self.side_streams
        return self._branch_streams[name] if on else torch.cuda.current_stream(self.device)

    def _grouped(self, Mp: int, d: int) -> bool:
        """A block's four weight gradients as ONE grouped launch (and its Linears' backward launches input gradients only)?  From
        ``wgrad_group_min`` padded rows x width up — and only where that launch runs BESIDE the chain: on one queue the paired
        launches (input + weight gradient together) are fewer nodes for the same work (round 6, batch 8 on one queue: 4.90 ms
        grouped, 4.80 paired).  Decided by the workspace's shape alone (not by ``_ddp_active``: buffers are sized by it)."""
        if Mp * d < self.wgrad_group_min:
            return False
        if self.side_streams is None:
            return getattr(self, 'Md', 0) >= self.side_min_rows
        return 'wside' in self.side_streams

    def forked_branches(self):
        """Names of the branch streams the step forks at the current workspace (empty: one chain on the current stream)."""
        cur = torch.cuda.current_stream(self.device)
        return sorted(n for n in self._branch_streams if self._branch(n) != cur)

    def _branch_prop(name):
        return property(lambda self: self._branch(name), lambda self, st: self._branch_streams.__setitem__(name, st))

    side, wside, pside, oside = _branch_prop('side'), _branch_prop('wside'), _branch_prop('pside'), _branch_prop('oside')
    del _branch_prop

    adamw_acc_gate = os.environ.get('VITAE_ADAMW_ACC_GATE', '1') != '0'
    wgrad_group_min = int(float(os.environ.get('VITAE_WGRAD_GROUP_MIN', '0.6e6')))
    bn_split_min_rows = int(os.environ.get('VITAE_BN_SPLIT_MIN_ROWS', '128'))   # predictor BatchNorm: rows per view from which the row-split kernels run
    target_one_pass = os.environ.get('VITAE_TARGET_ONE_PASS', '1') != '0'
    # optional explicit ascending block boundaries, e.g. "0,2,7,12" (uneven chunks: a smaller last, exposed bucket)
    enc_cuts = [int(v) for v in os.environ['VITAE_ENC_CUTS'].split(',')] if os.environ.get('VITAE_ENC_CUTS') else None

    def set_backward_chunks(self, n: int):
        """Number of encoder-backward phases.  More phases = smaller gradient buckets = an earlier start and a shorter
        exposed tail for the data-parallel all-reduce (each phase is its own graph replay)."""
        self.enc_chunks = max(1, min(int(n), self.cfg.depth))

    # decoder backward in one phase (default) or two (data parallel: the gradients of decoder_pred, the top half of the decoder
    # blocks and the predictor form a bucket of their own, whose all-reduce starts ~0.35 ms earlier)
    dec_chunks = 1

    @property
    def dec_cut(self) -> int:
        """first decoder block of the 'top' part (0 = everything in one phase)"""
        return self.cfg.decoder_depth // 2 if self.dec_chunks == 2 else 0

    def set_decoder_chunks(self, n: int):
        self.dec_chunks = 2 if (int(n) >= 2 and self.cfg.decoder_depth >= 2) else 1

    @property
    def N_PHASES(self) -> int:
        return self.dec_chunks + self.enc_chunks + 1

    def enc_chunk_bounds(self):
        """[(hi, lo)] block ranges of the encoder-backward phases, last block first; sizes differ by at most one."""
        d, n = self.cfg.depth, self.enc_chunks
        cuts = [round(d * i / n) for i in range(n + 1)]              # ascending block boundaries
        ec = getattr(self, 'enc_cuts', None)
        if ec and ec[0] == 0 and ec[-1] == d and len(ec) == n + 1:
            cuts = ec
        return [(cuts[i + 1] - 1, cuts[i]) for i in reversed(range(n))]

    def train_phase(self, k: int, view1, view2, noise, mask_ratio: float, update: bool = True,
                    accumulate: bool = False):
        """Phase k of one optimisation step (only kernel launches, no host sync):
        0 (.. dec_chunks - 1) = forward + losses + backward through decoder/predictor; then enc_chunks phases of encoder backward,
        top chunk first (the last one also does the patch embedding); the last phase = grad-norm + AdamW.  Gradient bucket k (ddp) is
        final after phase k.  Loss multipliers and lr must already be in ``hp``."""
        cfg = self.cfg
        n, nd = self.enc_chunks, self.dec_chunks
        self._epi_norm_on = bool(self.epi_norm and update and self._optimizer_in_backward_ok())
        if self._epi_norm_on and os.environ.get('VITAE_SQ_SPREAD', '1') == '0':      # (A/B knob: everything on acc[GRADSQ], the form up to round 5)
            lib.vitae_gemm_glds_set_wgrad_sqnorm(self.acc.data_ptr() + 8 * _C['VITAE_ACC_GRADSQ'])
        elif self._epi_norm_on:    # (spread slots: same-address double atomics retire one per ~10 ns — 576 of them per weight-gradient launch)
            lib.vitae_gemm_glds_set_wgrad_sqnorm_spread(self.acc.data_ptr() + 8 * _C['VITAE_ACC_SQ_BASE'], _C['VITAE_ACC_SQ_SLOTS'],
                                                        _C['VITAE_ACC_SQ_STRIDE'])
        else:
            lib.vitae_gemm_glds_set_wgrad_sqnorm(None)
        try:
            self._train_phase(k, view1, view2, noise, mask_ratio, update, accumulate)
        finally:
            lib.vitae_gemm_glds_set_wgrad_sqnorm(None)

    def _train_phase(self, k, view1, view2, noise, mask_ratio, update, accumulate):
        cfg = self.cfg
        n, nd = self.enc_chunks, self.dec_chunks
        if k == 0:
            # the zeroing of the token / vector gradient segment and the loss finalisation are not on the dependent chain:
            # the first goes in front of the forward, the second behind the decoder backward (12 us between the loss kernels)
            self.step_prologue(noise, accumulate)
            self.forward(view1, view2, noise, mask_ratio, training=True, defer_predictor_join=True, defer_finalize=True,
                         loss_with_grad=True)
            if cfg.contrastive:
                self.contrastive_loss_fwd()
                self.contrastive_loss_bwd()
            self.backward_dec(have_dp=cfg.contrastive, part='top' if nd == 2 else None)
            self.loss_finalize()
            if update and self._optimizer_in_backward_ok():
                self._opt_bucket(0)
        elif k == 1 and nd == 2:
            self.backward_dec(have_dp=cfg.contrastive, part='bottom')
            if update and self._optimizer_in_backward_ok():
                self._opt_bucket(1)
        elif nd <= k < nd + n:
            hi, lo = self.enc_chunk_bounds()[k - nd]
            self.backward_enc(hi, lo)
            if k == nd + n - 1:
                self.backward_tail()
            if update and self._optimizer_in_backward_ok():
                self._opt_bucket(k)
        elif k == nd + n and update:
            if self._optimizer_in_backward_ok() or self._ddp_bucket_opt:
                self._opt_tail()             # the buckets' matrices were stepped beside the backward
            else:
                self.grad_norm_and_step()
        if k == nd + n:
            self.step_epilogue()             # every micro-step, update or not: the device's ring position follows the host's

    def train_step_launch(self, view1, view2, noise, mask_ratio: float, update: bool = True, accumulate: bool = False):
        for k in range(self.N_PHASES):
            self.train_phase(k, view1, view2, noise, mask_ratio, update, accumulate)
        self.end_step_host()

    def set_loss_weights(self, edge_map_weight: float, contr_weight: float, accum_iter: int = 1, world_size: int = 1):
        """Loss weights + upstream gradient multipliers; 1/(accum_iter*world_size) makes a SUM
        all-reduce of the per-rank gradients their mean."""
        s = 1.0 / (accum_iter * world_size)
        self.set_hparams(edge_w=edge_map_weight, contr_w=contr_weight, g_recon=s, g_edge=edge_map_weight * s,
                         g_contr=contr_weight * s)
