"""3-D ViT masked autoencoder (+ SimSiam-style contrastive head) with the reference's Python surface
(reference: model/vit_autoenc.py:14-315) on top of the HIP engine.

Same class names, constructor keywords, sub-module / state-dict names and return tuples as the
reference, so ``utils.train_one_epoch.train_one_stage_epoch`` and the BraTS/EGD scripts drive it
unchanged.  ``forward`` is ONE autograd node: its forward runs ``HipMAEEngine.forward`` (kernels of
libvitae_hip.so), its backward runs ``HipMAEEngine.backward`` and publishes the gradients as
``param.grad`` views of the engine's flat gradient arena.  There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import os

from collections import OrderedDict
from functools import partial
from typing import Optional

import torch
from torch import nn

from .._abi import VitaeError
from ..engine import HP, HipMAEEngine, MAEConfig
from .model_utils.perceptual_loss import vgg_perceptual_loss
from .model_utils.sobel_filter import SobelFilter3d
from .model_utils.vit_helpers import get_3d_sincos_pos_embed
from .vit import Block, PatchEmbed3D


class _NoPerceptualLoss(nn.Module):
    """perceptual_weight == 0: the term is zero whatever the network (reference vit_autoenc.py:229-230)."""

    def forward(self, pred_vol, target_vol):
        return torch.zeros((), device=pred_vol.device)


class _MAEStep(torch.autograd.Function):
    """(anchor, model, view1, view2, noise, mask_ratio, edge_w) ->
    (losses[5], pred, mask, p1, p2).  Gradients flow to the parameters as a side effect
    (param.grad = view of the engine's gradient arena); the anchor only keeps the node alive."""

    @staticmethod
    def forward(ctx, anchor, model, view1, view2, noise, mask_ratio, edge_w):
        eng: HipMAEEngine = model._engine
        eng.set_hparams(edge_w=edge_w)
        eng.forward(view1, view2, noise, mask_ratio, training=model.training)
        ctx.model, ctx.edge_w = model, float(edge_w)
        ctx.set_materialize_grads(False)
        b, cfg, B = eng.buf, eng.cfg, view1.shape[0]
        losses = eng.losses[:4].clone()
        if model.perceptual_weight:
            # no-gradient logging term on (unpatchified prediction, target volume) — reference vit_autoenc.py:228-231;
            # unpatchify(patchify(x)) == x, so the target volume is the input itself
            percep = model.perceptual_weight * model.perceptual_loss(b['pred_vol'], view1)
            losses[3] = percep
            losses[0] = losses[0] + percep
        pred = b['predfull'][:, 1:, :]
        mask = b['mask'][:B]
        if cfg.contrastive:
            R = eng.R
            p1, p2 = b['pout'][:R], b['pout'][R:]
        else:
            p1 = p2 = None
        ctx.mark_non_differentiable(mask)
        return losses, pred, mask, p1, p2

    @staticmethod
    def backward(ctx, g_losses, g_pred, g_mask, g_p1, g_p2):
        model = ctx.model
        eng: HipMAEEngine = model._engine
        if g_pred is not None:
            raise VitaeError('gradients through `pred` other than via the returned losses are not supported')
        hp = eng.hp
        if g_losses is None:
            hp[HP['G_RECON']] = 0.0
            hp[HP['G_EDGE']] = 0.0
        else:   # total = w_e*edge + recon (+0): d/d recon = g0 + g2 ; d/d raw_edge = w_e*g0 + g1
            hp[HP['G_RECON']] = g_losses[0] + g_losses[2]
            hp[HP['G_EDGE']] = ctx.edge_w * g_losses[0] + g_losses[1]
        have_dp = eng.cfg.contrastive and (g_p1 is not None or g_p2 is not None)
        if have_dp:
            R = eng.R
            dp = eng.buf['dp']
            eng._dp16_ready = False      # the gradient arrives in fp32 here: the predictor's backward makes its own bf16 copy
            for half, g in ((dp[:R], g_p1), (dp[R:], g_p2)):
                if g is None:
                    half.zero_()
                else:
                    half.copy_(g)
        fresh = all(p.grad is None for p in model._trainable)
        eng.begin_grad_window(accumulate=not fresh)
        eng.backward(have_dp)
        if fresh:
            for n, p in model._trainable_named:
                p.grad = eng.g[n]
        return None, None, None, None, None, None, None


_SLOTS = int(os.environ.get('VITAE_INPUT_SLOTS', '2'))     # host batches: 2 = double-buffered on a copy stream, 1 = one slot, 0 = main stream
_DOUBLE_BUFFER = _SLOTS >= 2
_MAX_DIRECT = int(os.environ.get('VITAE_DIRECT_GRAPHS', '4'))   # device batches read in place: graphs kept per runner (0 = always stage)


class _InputSlot:
    """One set of static input buffers of the step graph + the events that order its refills."""

    def __init__(self, cfg, B, dev):
        self.v1 = torch.empty(B, cfg.in_chans, *cfg.volume_size, dtype=torch.float32, device=dev)
        self.v2 = torch.empty_like(self.v1) if cfg.contrastive else None
        self.noise = torch.empty((2 * B if cfg.contrastive else B), cfg.num_patches, dtype=torch.float32, device=dev)
        self.loaded = torch.cuda.Event()
        self.free = None          # recorded after the last step that read this slot


class _DirectSlot:
    """A device-resident batch read in place by a graph captured for its addresses (no staging copy)."""

    def __init__(self, v1, v2, noise):
        self.v1, self.v2, self.noise = v1, v2, noise
        self.loaded = torch.cuda.Event()
        self.free = None


class _StepRunner:
    """One (batch size, mask ratio, update?, accumulate?) variant of the fused optimisation step:
    static input buffers + either eager launches or a captured HIP graph of the whole step.

    Host batches are double-buffered: ``load`` fills the slot the NEXT ``run`` will read on a copy stream, so the
    transfer of batch i+1 (113 MB of host->device traffic for two 96^3 x 4ch views at batch 4, ~2 ms over PCIe)
    overlaps step i instead of preceding step i+1; there is one captured graph (set) per slot.

    Device batches are staged into a slot in program order (226 MB of HBM traffic, ~0.1 ms of a 5.6 ms step at batch 4)
    — unless the same tensors come back (a dataset that lives in HBM, or the caching allocator handing the loader the
    same blocks again): from the second time on such a batch gets a graph captured on its own addresses and is read in
    place.  At most ``VITAE_DIRECT_GRAPHS`` of those are kept per runner."""

    def __init__(self, model, B, mask_ratio, update, accumulate, use_graph):
        self.model, self.eng = model, model._engine
        self.B, self.mask_ratio, self.update, self.accumulate = B, mask_ratio, update, accumulate
        eng, cfg, dev = self.eng, model._cfg, model._engine.device
        st = model._static.get(B)
        if st is None:
            st = model._static[B] = {'slots': [_InputSlot(cfg, B, dev), _InputSlot(cfg, B, dev)], 'next': 0, 'cur': 0,
                                     'copy': torch.cuda.Stream(device=dev)}
        self.st = st
        self.graphs = {}              # slot index, or ('direct', ptr1, ptr2) -> captured graph(s)
        self.use_graph = use_graph
        self._direct, self._seen = None, {}
        self._ws_gen = self.eng.ws_gen

    @property
    def slot(self):
        return self._direct if self._direct is not None else self.st['slots'][self.st['cur']]

    @property
    def _gkey(self):
        d = self._direct
        return self.st['cur'] if d is None else ('direct', d.v1.data_ptr(), 0 if d.v2 is None else d.v2.data_ptr())

    def _direct_key(self, view1, view2):
        """Graph key of a device batch that could be read in place, or None (needs the staging copy)."""
        ref = self.st['slots'][0]
        if not self.use_graph or _MAX_DIRECT <= 0 or (ref.v2 is None) != (view2 is None):
            return None
        for v in (view1,) if view2 is None else (view1, view2):
            if not (v.is_cuda and v.device == ref.v1.device and v.dtype == torch.float32 and v.is_contiguous()
                    and v.shape == ref.v1.shape):
                return None
        return ('direct', view1.data_ptr(), 0 if view2 is None else view2.data_ptr())

    # the buffers the next run() reads (kept as attributes for callers that fill them in place)
    v1 = property(lambda self: self.slot.v1)
    v2 = property(lambda self: self.slot.v2)
    noise = property(lambda self: self.slot.noise)

    def load(self, view1, view2, ready: bool = False):
        """Stage one batch and fresh masking noise for the next ``run``.
        * HOST tensors (pinned for a truly asynchronous transfer): copied on the copy stream into the slot the next
          ``run`` reads, alternating between two slots, so the host-to-device transfer of batch i+1 overlaps step i.
        * DEVICE tensors: copied on the current stream into the current slot — measured: a device-to-device copy
          that overlaps the step gains nothing (it competes for HBM with the kernels: 5.82 vs 5.81 ms) and costs
          0.35 ms in the data-parallel step, so it stays in program order.  ``ready``: kept for callers (no effect)."""
        st = self.st
        main = torch.cuda.current_stream(self.eng.device)
        host = not view1.is_cuda and (view2 is None or not view2.is_cuda) and _SLOTS >= 1
        if host:
            s = st['next']
            st['cur'], st['next'] = s, (s ^ 1) if _DOUBLE_BUFFER else s
            slot, copy = st['slots'][s], st['copy']
            if slot.free is not None:
                copy.wait_event(slot.free)             # the step that last read this slot is done with it
        else:
            slot, copy = st['slots'][st['cur']], main
        m = self.model
        self._direct = None
        key = None if host else self._direct_key(view1, view2)
        if key is not None:
            direct = key in self.graphs
            if not direct:
                if len(self._seen) > 64:
                    self._seen.clear()
                self._seen[key] = self._seen.get(key, 0) + 1
                direct = self._seen[key] >= 2 and sum(isinstance(k, tuple) for k in self.graphs) < _MAX_DIRECT
            if direct:
                slot = self._direct = _DirectSlot(view1, view2, st['slots'][0].noise)
                self._stage_noise(slot)
                slot.loaded.record(main)
                return
        with torch.cuda.stream(copy):
            slot.v1.copy_(view1, non_blocking=True)
            if slot.v2 is not None:
                slot.v2.copy_(view2, non_blocking=True)
            self._stage_noise(slot)
            slot.loaded.record(copy)

    def _stage_noise(self, slot):
        """The torch.rand of vit_autoenc.py:139.  Fused step: the masking noise is drawn by the step's own first launch
        (vitae_step_prologue: Philox keyed by the engine's seed and the step number) — nothing is launched here; injected noise
        (``set_masking_noise``: parity tests) is copied into the slot and the step is told to keep it."""
        m, eng = self.model, self.eng
        if m._noise_queue:
            slot.noise.copy_(m._draw_noise(2 if m._contrastive else 1, self.B, slot.noise.device))
            eng.set_hparams(noise_keep=1.0)
        else:
            eng.set_hparams(noise_keep=0.0)

    def _phase(self, k):
        sl = self.slot
        self.eng.train_phase(k, sl.v1, sl.v2, sl.noise, self.mask_ratio, update=self.update, accumulate=self.accumulate)

    def _exchange(self):
        red = self.model._reducer
        return red is not None and red.active and self.update

    def _after_phase(self, i):
        """Data parallel: bucket i is final after phase i — launch its all-reduce; once it has landed its matrices are stepped
        on the optimiser stream underneath the next backward phases (the reduced values are read from the wire buffer when it
        is bf16); after the last backward phase the token / vector tail goes and everything is joined."""
        eng, red = self.eng, self.model._reducer
        nph = eng.N_PHASES
        if i >= nph - 1:
            return
        bucket_opt = eng._ddp_bucket_opt
        red.launch(i)
        if bucket_opt:
            with torch.cuda.stream(eng.oside):
                red.wait_bucket(i, copy_back=eng.grads_wire16 is None)
            eng._opt_bucket(i, wait_main=False)
        if i == nph - 2:
            red.launch(nph - 1)       # tokens + vectors
            red.wait(copy_back=eng.grads_wire16 is None)
            if bucket_opt:
                torch.cuda.current_stream(eng.device).wait_stream(eng.oside)

    def _run_group(self, grp, inside):
        for k in grp:
            self._phase(k)
            if inside:
                self._after_phase(k)

    def _capture(self, groups, inside):
        """Warm up once (loads code objects, sizes the workspace; state restored afterwards), then
        capture each group of phases as one HIP graph.  ``inside``: the gradient exchange is part of the launch list (the
        RCCL C-ABI reducer), i.e. captured with it."""
        eng = self.eng
        keep = [eng.params, eng.grads] + ([eng.opt_state['exp_avg'], eng.opt_state['exp_avg_sq']] if eng.opt_state else [])
        keep += [t for k, t in eng.buffers.items() if k.startswith('predictor.1.')]
        keep.append(eng.hp)                    # (the device-side count of applied AdamW steps lives in it)
        snap = [t.clone() for t in keep]
        step = eng.opt_step
        cur = torch.cuda.current_stream(eng.device)
        # (a high-priority capture stream for the main chain was tried so that the side branches only fill free CUs:
        # the whole step ran 2x slower, 11.9 ms — HIP's high-priority queue is not a free lunch here)
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._run_group(range(eng.N_PHASES), inside)
        cur.wait_stream(side)
        eng.end_step_host()                    # the warm-up was a launched step: the ring position moved on both sides
        for t, s in zip(keep, snap):
            t.copy_(s)
        eng.opt_step = step
        eng.refresh_shadow(force=True)     # restoring bumped the arena's version: re-cast now, not inside the graph
        torch.cuda.synchronize(eng.device)
        graphs = []
        for grp in groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                self._run_group(grp, inside)
            graphs.append(g)
        self.graphs[self._gkey] = graphs

    def run(self):
        """Enqueue one optimisation step.  Single process: one graph (or one eager launch list).
        Data parallel: every backward phase is followed by the asynchronous all-reduce of the gradient bucket
        it completes, then wait + grad-norm/AdamW (no exchange on gradient-accumulation micro-steps).  With torch's
        process group the collectives are host-issued between per-phase graphs; with the RCCL C-ABI reducer they are nodes
        of ONE graph."""
        eng, red = self.eng, self.model._reducer
        exchange = self._exchange()
        inside = exchange and getattr(red, 'native', False)
        nph = eng.N_PHASES
        groups = [[k] for k in range(nph)] if (exchange and not inside) else [list(range(nph))]
        eng._wire_ready = exchange and eng.grads_wire16 is not None    # read at launch / capture time of the last phase
        eng._ddp_active = exchange          # buckets change after their phase (all-reduce): the runner steps them below
        bucket_opt = exchange and eng.overlap_optimizer and eng.opt_state is not None
        eng._ddp_bucket_opt = bucket_opt    # last phase = tokens/vectors + norm only
        sl = self.slot
        main = torch.cuda.current_stream(eng.device)
        main.wait_event(sl.loaded)                     # this slot's batch has landed
        if self._ws_gen != eng.ws_gen:                 # the engine freed a workspace some captured graph may point into
            self.graphs.clear()
            self._ws_gen = eng.ws_gen
        eng._alloc(self.B, self.mask_ratio)            # this runner's workspace is the current one (replays bypass forward())
        if self.use_graph and self.graphs.get(self._gkey) is None:
            self._capture(groups, inside)
        graphs = self.graphs.get(self._gkey)
        for i, grp in enumerate(groups):
            if self.use_graph:
                graphs[i].replay()
            else:
                self._run_group(grp, inside)
            if exchange and not inside:
                self._after_phase(i)
        eng.end_step_host()
        sl.free = torch.cuda.Event()
        sl.free.record(main)


class MaskedAutoencoderViT(nn.Module):
    """Masked Autoencoder with a 3-D VisionTransformer backbone (reference vit_autoenc.py:14-238)."""

    _contrastive = False

    def __init__(self, volume_size=224, patch_size=16, in_chans=3,
                 embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, args=None, precision=None):
        super().__init__()
        if norm_pix_loss:
            raise VitaeError('norm_pix_loss=True is not on the HIP path (never enabled by model_factory.get_models)')
        self.patch_embed = PatchEmbed3D(volume_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.embed_dim = embed_dim
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True,
                                                   norm_layer=norm_layer) for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        p = self.patch_embed.patch_size[0]
        self.decoder_pred = nn.Linear(decoder_embed_dim, p ** 3 * in_chans, bias=True)
        self.sobel_filter3D = SobelFilter3d()
        self.args = args
        self.perceptual_weight = 1 if args is None else args.perceptual_weight
        # reference vit_autoenc.py:55-58.  With weight 0 (the shipped configuration) the term is identically zero and the
        # hook carries no VGG tensors, so state-dict keys equal SURVEY A.5; with a weight, the VGG16 feature stack is built
        # (keys perceptual_loss.slice*.*, like the real reference) and the caller loads its weights — neither torchvision's
        # nor the reference's ckp-399.pth exist here (parity with those weights unpinned).
        self.perceptual_loss = (vgg_perceptual_loss(use_imagenet=getattr(args, 'use_imagenet', False)) if self.perceptual_weight
                                else _NoPerceptualLoss())
        print(f"Using perceptual weight of {self.perceptual_weight}")
        self.norm_pix_loss = norm_pix_loss
        eps = getattr(self.norm, 'eps', 1e-6)
        self._cfg = MAEConfig(volume_size=self.patch_embed.volume_size, patch_size=p, in_chans=in_chans,
                              embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                              decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth,
                              decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio,
                              contrastive=self._contrastive, ln_eps=eps)
        self._precision = precision or getattr(args, 'precision', None) or 'fp32'
        self._engine: Optional[HipMAEEngine] = None
        self._noise_queue = []
        self._static, self._runners = {}, {}
        self._reducer = None
        self.initialize_weights()

    # ------------------------------------------------------------------ init (vit_autoenc.py:65-98)
    def initialize_weights(self):
        grid = self.patch_embed.grid_size
        with torch.no_grad():
            for name, dim in (('pos_embed', self.pos_embed.shape[-1]), ('decoder_pos_embed', self.decoder_pos_embed.shape[-1])):
                table = get_3d_sincos_pos_embed(dim, grid, cls_token=True)
                getattr(self, name).copy_(torch.from_numpy(table).float().unsqueeze(0))
            w = self.patch_embed.proj.weight
            nn.init.xavier_uniform_(w.view(w.shape[0], -1))      # like nn.Linear, not like a conv
            nn.init.normal_(self.cls_token, std=.02)
            nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------ engine plumbing
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        eng = self.__dict__.get('_engine')
        if eng is not None and any(p.data_ptr() != eng.p[n].data_ptr() or p.device != eng.device
                                   for n, p in self._trainable_named):
            self._engine = None   # parameters moved / were re-typed: re-adopt them into a fresh arena lazily
        return out

    def _ensure_engine(self, device: torch.device) -> HipMAEEngine:
        if device.type != 'cuda':
            raise VitaeError(f'input is on {device}: vit_ae_plus_plus_amd computes on MI355X only — there is '
                             f'no CPU fallback (the CPU restatement lives in oracle/ for tests)')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self._engine is not None and self._engine.device == device:
            return self._engine
        named = OrderedDict((n, p) for n, p in self.named_parameters() if p.requires_grad)
        bufs = {'pos_embed': self.pos_embed.data.to(device).reshape(-1, self.pos_embed.shape[-1]).contiguous(),
                'decoder_pos_embed': self.decoder_pos_embed.data.to(device).reshape(
                    -1, self.decoder_pos_embed.shape[-1]).contiguous()}
        if self._contrastive:
            bn = self.predictor[1]
            for k in ('running_mean', 'running_var', 'num_batches_tracked'):
                t = getattr(bn, k)
                if t.device != device:
                    setattr(bn, k, t.to(device))
                bufs['predictor.1.' + k] = getattr(bn, k)
        eng = HipMAEEngine(self._cfg, named, bufs, device, precision=self._precision)
        for n, p in named.items():          # parameters become views of the flat arena
            p.data = eng.p[n]
            p.grad = None
        self._trainable_named = list(named.items())
        self._trainable = [p for _, p in self._trainable_named]
        self._engine = eng
        return eng

    def _step_runner(self, B, mask_ratio, update, accumulate, use_graph) -> _StepRunner:
        key = (int(B), float(mask_ratio), bool(update), bool(accumulate), bool(use_graph), id(self._engine))
        r = self._runners.get(key)
        if r is None:
            r = self._runners[key] = _StepRunner(self, int(B), float(mask_ratio), bool(update), bool(accumulate),
                                                 bool(use_graph))
        return r

    def load_state_dict(self, state_dict, strict=True, **kw):
        # tolerate the real reference's extra VGG tensors (perceptual_loss.*) when the hook is off, SURVEY §8b
        own = any(k.startswith('perceptual_loss.') for k in self.state_dict().keys())
        sd = {k: v for k, v in state_dict.items() if own or not k.startswith('perceptual_loss.')}
        out = super().load_state_dict(sd, strict=strict, **kw)
        eng = self._engine
        if eng is not None:   # frozen tables are engine-side copies
            eng.buffers['pos_embed'].copy_(self.pos_embed.data.reshape(eng.buffers['pos_embed'].shape))
            eng.buffers['decoder_pos_embed'].copy_(
                self.decoder_pos_embed.data.reshape(eng.buffers['decoder_pos_embed'].shape))
        return out

    @staticmethod
    def _set_exchange_buckets(eng, enc_chunks):
        """Encoder gradient buckets of the data-parallel step.  Explicit ``enc_chunks``: that many, even.  Otherwise (and unless
        VITAE_ENC_CHUNKS / VITAE_ENC_CUTS say something else): four, the LAST one only block 0 + the patch embedding — it is the
        bucket whose all-reduce nothing can hide, so it is the small one (ring all-reduce modelled on one GPU at 150 / 300 GB/s
        bus bandwidth, tools/probes/r2_ddp_model.sh: 5.63 / 5.07 ms against 5.80 / 5.27 ms with three even buckets)."""
        if enc_chunks is not None:
            eng.set_backward_chunks(enc_chunks)
            eng.set_decoder_chunks(1)
            return
        # decoder: optionally two buckets (VITAE_DEC_CHUNKS=2: decoder_pred + top blocks + predictor first, its all-reduce starts
        # ~0.35 ms earlier).  Modelled ring (tools/probes/r2_ddp_model4.sh): 5.64 vs 5.78 ms at 150 GB/s bus bandwidth, but 5.26 vs
        # 5.22 at 200 and 5.14 vs 5.11 at 300 (one more graph replay and bucket) — off unless the links turn out slow
        eng.set_decoder_chunks(int(os.environ.get('VITAE_DEC_CHUNKS', '1')))
        if os.environ.get('VITAE_ENC_CHUNKS') or os.environ.get('VITAE_ENC_CUTS'):
            return
        d = eng.cfg.depth
        n = min(4, d)
        eng.set_backward_chunks(n)
        eng.enc_cuts = [0, d] if n == 1 else [0] + [1 + (d - 1) * i // (n - 1) for i in range(n)]

    def enable_data_parallel(self, device=None, group=None, force=False, comm_dtype=None, enc_chunks=None, native=None):
        """One process per GPU: broadcast rank 0's replica and all-reduce gradient buckets over RCCL
        (overlapped with backward) inside the fused step.  No-op for a single process.
        ``comm_dtype=torch.bfloat16`` sends the gradients rounded to bf16 (half the bytes on xGMI);
        ``enc_chunks`` = number of (even) encoder gradient buckets (default: ``_set_exchange_buckets``).  ``native`` (default: environment VITAE_DDP_NATIVE=1): exchange
        through the C ABI (``vitae_ddp_*``: RCCL on a side HIP stream, captured inside the step graph) instead of through
        torch.distributed's process group (collectives issued by the host between per-phase graphs)."""
        from .. import ddp
        if native is None:
            native = os.environ.get('VITAE_DDP_NATIVE', '0') == '1'
        self.disable_data_parallel()        # a previous reducer's communicator / captured graphs go first
        if self._engine is not None:
            self._engine._noise_seed_now()  # the rank may have become known since the engine was built: new masking-noise key,
            self._runners.clear()           # ... and no graph captured with the old one survives
        if native and (ddp.is_distributed() or force):
            eng = self._ensure_engine(torch.device(device) if device is not None else next(self.parameters()).device)
            ddp.broadcast_parameters(eng, 0, group)
            self._set_exchange_buckets(eng, enc_chunks)
            self._reducer = ddp.RcclBucketReducer(eng.grads, ddp.engine_bucket_ranges(eng), eng.device, comm_dtype=comm_dtype,
                                                  force=force, group=group)
            wired = self._reducer.wire is not None and self._reducer.active
            if wired:
                eng.grads_wire16 = self._reducer.wire
                self._reducer.cast_ranges = eng.wire_uncovered_ranges()
            eng.grads_wire16 = self._reducer.wire if wired else None
            self._runners.clear()
            return self._reducer
        if not ddp.is_distributed() and not (force and torch.distributed.is_initialized()):
            self._reducer = None
            if self._engine is not None:
                self._engine.grads_wire16 = None
            return None
        eng = self._ensure_engine(torch.device(device) if device is not None else next(self.parameters()).device)
        ddp.broadcast_parameters(eng, 0, group)
        self._set_exchange_buckets(eng, enc_chunks)
        self._reducer = ddp.GradBucketReducer(eng.grads, ddp.engine_bucket_ranges(eng), group=group, force=force,
                                              comm_dtype=comm_dtype)
        if eng.device.type == 'cuda' and self._reducer.active and os.environ.get('VITAE_DDP_PICK_STREAMS', '1') != '0':
            # the all-reduce kernels and the per-bucket AdamW must not share a hardware queue with the backward (or with each
            # other): measured once, here (ddp.pick_streams)
            try:
                eng.oside, self._reducer.comm_stream, self._stream_report = ddp.pick_streams(eng.device)
            except Exception as e:   # pragma: no cover - never let a tuning probe cost the run
                self._stream_report = {'ok': False, 'error': repr(e)[:200]}
        if self._reducer.wire is not None and self._reducer.active:
            eng.grads_wire16 = self._reducer.wire
            self._reducer.cast_ranges = eng.wire_uncovered_ranges()   # the rest is written by the wgrad epilogues
        # bf16 exchange: the fused grad-norm + AdamW read the reduced gradients straight from the wire buffer
        eng.grads_wire16 = self._reducer.wire if (self._reducer.wire is not None and self._reducer.active) else None
        self._runners.clear()
        return self._reducer

    def disable_data_parallel(self):
        """Drop the gradient exchange (and every step graph captured with it); a native reducer gives RCCL's communicator back."""
        red = self._reducer
        if red is not None:
            self._runners.clear()
            if getattr(red, 'native', False):
                # the step graphs hold RCCL nodes of this communicator: they must be GONE (not merely unreferenced) before it is —
                # a graph destroyed later, next to a capture that uses the next communicator, crashed hipGraphLaunch (seen with
                # two graph-route data-parallel models in one process)
                import gc
                gc.collect()
                torch.cuda.synchronize()
                red.close()
        self._reducer = None
        if self._engine is not None:
            self._engine.grads_wire16 = None

    @property
    def engine(self) -> Optional[HipMAEEngine]:
        return self._engine

    def set_precision(self, precision: str):
        """'fp32' (exact-fp32 MFMA, the reference's precision), 'fp32x3' (fp32 operands split into bf16 hi + lo inside the GEMMs,
        three bf16 MFMAs per product: fp32-grade results at several times the rate) or 'bf16' (bf16 MFMA, fp32 accumulate)."""
        if precision not in ('fp32', 'fp32x3', 'bf16'):
            raise ValueError(f"precision must be 'fp32', 'fp32x3' or 'bf16', not {precision!r}")
        self._precision = precision
        self._engine = None

    def set_masking_noise(self, *noises):
        """Queue [B, L] noise tensors consumed (in order) instead of drawing torch.rand
        (vit_autoenc.py:139) — the parity hook of SURVEY §7.2 'RNG'."""
        self._noise_queue = [n for n in noises]

    def _draw_noise(self, n_views, B, device):
        L = self.patch_embed.num_patches
        parts = []
        for _ in range(n_views):
            if self._noise_queue:
                z = self._noise_queue.pop(0).to(device=device, dtype=torch.float32)
                assert tuple(z.shape) == (B, L)
            else:
                z = torch.rand(B, L, device=device)
            parts.append(z)
        return parts[0].contiguous() if n_views == 1 else torch.cat(parts, 0)

    def _prep(self, v):
        if v.dtype != torch.float32 or not v.is_contiguous():
            v = v.contiguous().float()
        return v

    # ------------------------------------------------------------------ pure index permutations
    def patchify(self, volume):
        """[N, C, L, H, W] -> [N, l*h*w, p^3*C]  (vit_autoenc.py:100-113; order (r, p, q, c))."""
        p = self.patch_embed.patch_size[0]
        N, C = volume.shape[:2]
        l, h, w = (s // p for s in volume.shape[2:])
        assert all(s % p == 0 for s in volume.shape[2:])
        x = volume.reshape(N, C, l, p, h, p, w, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
        return x.reshape(N, l * h * w, p ** 3 * C)

    def unpatchify(self, x):
        """[N, l*h*w, p^3*C] -> [N, C, L, H, W]  (vit_autoenc.py:115-128)."""
        p = self.patch_embed.patch_size[0]
        l, h, w = self.patch_embed.grid_size
        assert l * h * w == x.shape[1]
        N = x.shape[0]
        x = x.reshape(N, l, h, w, p, p, p, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return x.reshape(N, -1, l * p, h * p, w * p)

    # ------------------------------------------------------------------ forward
    def _step(self, view1, view2, mask_ratio, edge_map_weight):
        view1 = self._prep(view1)
        view2 = self._prep(view2) if view2 is not None else None
        eng = self._ensure_engine(view1.device)
        noise = self._draw_noise(2 if self._contrastive else 1, view1.shape[0], view1.device)
        return _MAEStep.apply(self.cls_token, self, view1, view2, noise, float(mask_ratio), float(edge_map_weight))

    def forward(self, sample, mask_ratio=0.75, edge_map_weight=0):
        """-> ([loss, raw_edge_mse, recon, percep], pred [N, L, p^3 C], mask [N, L])  (vit_autoenc.py:234-238)"""
        losses, pred, mask, _, _ = self._step(sample, None, mask_ratio, edge_map_weight)
        return [losses[0], losses[1], losses[2], losses[3]], pred, mask


class ContrastiveMAEViT(MaskedAutoencoderViT):
    """MAE + SimSiam predictor on the two views' latents (reference vit_autoenc.py:241-285)."""

    _contrastive = True

    def __init__(self, volume_size=224, patch_size=16, in_chans=3,
                 embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, args=None, use_proj=False,
                 precision=None):
        if use_proj:
            raise VitaeError('use_proj=True (projection_head) is never used by the reference pre-training path')
        super().__init__(volume_size=volume_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                         depth=depth, num_heads=num_heads, decoder_embed_dim=decoder_embed_dim,
                         decoder_depth=decoder_depth, decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio,
                         norm_layer=norm_layer, norm_pix_loss=norm_pix_loss, args=args, precision=precision)
        self.use_proj = use_proj
        # built after the base init, exactly like the reference (default nn.Linear / BatchNorm1d init)
        self.predictor = nn.Sequential(nn.Linear(embed_dim, embed_dim, bias=False), nn.BatchNorm1d(embed_dim),
                                       nn.ReLU(inplace=True), nn.Linear(embed_dim, embed_dim))

    def forward(self, view1, view2, mask_ratio=0.75, edge_map_weight=0):
        """-> (loss_list, pred, mask, p1, p2, z1.detach(), z2.detach())  (vit_autoenc.py:270-285)"""
        losses, pred, mask, p1, p2 = self._step(view1, view2, mask_ratio, edge_map_weight)
        eng = self._engine
        R = eng.R
        z = eng.buf['latent']
        return [losses[0], losses[1], losses[2], losses[3]], pred, mask, p1, p2, z[:R].detach(), z[R:].detach()


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def contr_mae_vit_base_patch16_dec512d8b(**kwargs):
    return ContrastiveMAEViT(embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8,
                             decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def contr_mae_vit_large_patch16_dec512d8b(**kwargs):
    return ContrastiveMAEViT(embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                             decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def contr_mae_vit_tiny_patch16(**kwargs):
    """BASELINE config 1 (tiny plumbing model: D=128, depth 2, 4 heads; decoder 64 / 1 / 4)."""
    return ContrastiveMAEViT(embed_dim=128, depth=2, num_heads=4, decoder_embed_dim=64, decoder_depth=1,
                             decoder_num_heads=4, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_tiny_patch16(**kwargs):
    return MaskedAutoencoderViT(embed_dim=128, depth=2, num_heads=4, decoder_embed_dim=64, decoder_depth=1,
                                decoder_num_heads=4, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


# recommended archs (same aliases as the reference, vit_autoenc.py:313-315)
mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
contr_mae_vit_base_patch16 = contr_mae_vit_base_patch16_dec512d8b
contr_mae_vit_large_patch16 = contr_mae_vit_large_patch16_dec512d8b
