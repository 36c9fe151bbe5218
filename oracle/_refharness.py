"""Import the REAL reference (``/root/reference``) behind the three stubs of SURVEY Appendix A.

TEST INFRASTRUCTURE, build-container only: ``/root/reference`` does not exist on the GPU box, so
nothing that runs there (``-m gpu`` tests, ``smoke()``, ``bench.py``) may import this module.  It is
used by ``oracle/gen_golden.py`` (fixture generation) and by the optional CPU test
``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent).

Stubs (none of them is executed on the MAE path, they only let the imports succeed):
  1. ``timm.models.helpers.{named_apply, adapt_input_conv}``, ``timm.models.layers.weight_init``
     (imported by model/vit.py:8-9 and model/model_utils/vit_helpers.py:6; timm is not installed);
  2. ``model.model_utils.perceptual_loss.vgg_perceptual_loss`` -> zero (needs torchvision + the
     unshipped ckp-399.pth; exact for perceptual_weight=0);
  3. ``torch.cuda.synchronize`` / ``torch.cuda.empty_cache`` no-ops for the training loop on a
     GPU-less host (utils/train_one_epoch.py:76,105).
"""
from __future__ import annotations

import contextlib
import os
import sys
import types
from argparse import Namespace

import torch

REFERENCE_ROOT = os.environ.get('VITAE_REFERENCE_ROOT', '/root/reference')


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'model'))


_MODS = None


def import_reference():
    """Returns a namespace with the reference's modules (model_factory, vit_autoenc, vit,
    vit_helpers, sobel_filter, gaussian_filter, train_one_epoch, misc, lr_sched)."""
    global _MODS
    if _MODS is not None:
        return _MODS
    if not reference_available():
        raise RuntimeError(f'reference not found under {REFERENCE_ROOT}')
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def named_apply(fn, module, name='', depth_first=True, include_root=False):
        raise NotImplementedError('timm stub: not on the MAE path')

    def adapt_input_conv(in_chans, conv_weight):
        raise NotImplementedError('timm stub: not on the MAE path')

    def lecun_normal_(t):
        raise NotImplementedError('timm stub: not on the MAE path')

    mod('timm')
    mod('timm.models')
    mod('timm.models.helpers', named_apply=named_apply, adapt_input_conv=adapt_input_conv)
    mod('timm.models.layers')
    mod('timm.models.layers.weight_init', trunc_normal_=torch.nn.init.trunc_normal_,
        lecun_normal_=lecun_normal_)

    class vgg_perceptual_loss(torch.nn.Module):
        def __init__(self, requires_grad=False, use_imagenet=False):
            super().__init__()

        def forward(self, a, b):
            return torch.zeros(())

    # our repo must not shadow the reference's top-level ``model`` / ``utils`` packages
    for k in [k for k in sys.modules if k == 'model' or k.startswith('model.')
              or k == 'utils' or k.startswith('utils.')]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import model.model_utils  # noqa: F401  (package, so the stub below nests correctly)
        mod('model.model_utils.perceptual_loss', vgg_perceptual_loss=vgg_perceptual_loss)
        from model import model_factory, vit, vit_autoenc
        from model.model_utils import gaussian_filter, sobel_filter, vit_helpers
        from utils import lr_sched, misc, train_one_epoch
    finally:
        sys.path.remove(REFERENCE_ROOT)
    _MODS = Namespace(model_factory=model_factory, vit=vit, vit_autoenc=vit_autoenc,
                      vit_helpers=vit_helpers, sobel_filter=sobel_filter,
                      gaussian_filter=gaussian_filter, train_one_epoch=train_one_epoch,
                      misc=misc, lr_sched=lr_sched)
    return _MODS


def reference_args(**kw) -> Namespace:
    """The attributes the hot path reads off ``args``."""
    base = dict(use_imagenet=False, perceptual_weight=0, volume_size=96, in_channels=4,
                patch_size=16, model='mae_vit_base_patch16', accum_iter=1, mask_ratio=0.75,
                contr_weight=0.001, warmup_epochs=40, lr=1e-3, min_lr=0.0, epochs=50)
    base.update(kw)
    return Namespace(**base)


@contextlib.contextmanager
def injected_noise(noises):
    """Make the reference's ``torch.rand(N, L, device=...)`` (vit_autoenc.py:139) return the
    queued ``noises`` in order, so masks are identical on both sides."""
    queue = list(noises)
    real = torch.rand

    def fake(*size, **kw):
        if kw.get('generator') is None and queue and tuple(queue[0].shape) == tuple(
                size[0] if len(size) == 1 and not isinstance(size[0], int) else size):
            return queue.pop(0).clone()
        return real(*size, **kw)

    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = real
    assert not queue, f'{len(queue)} injected noise tensors were not consumed'


@contextlib.contextmanager
def cpu_training_loop_patches():
    """Stub 3: lets utils/train_one_epoch.py run on a host without a GPU."""
    sync, empty = torch.cuda.synchronize, torch.cuda.empty_cache
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    try:
        yield
    finally:
        torch.cuda.synchronize, torch.cuda.empty_cache = sync, empty


def build_reference_model(cfg, factory_args=None):
    """Construct the reference module for an ``oracle.mae_ref.RefConfig`` (direct ctor call, the
    same call ``model_factory.get_models`` makes at model/model_factory.py:12)."""
    from functools import partial
    ref = import_reference()
    cls = ref.vit_autoenc.ContrastiveMAEViT if cfg.contrastive else ref.vit_autoenc.MaskedAutoencoderViT
    args = factory_args or reference_args()
    vol = cfg.volume_size if len(set(cfg.volume_size)) > 1 else cfg.volume_size[0]
    return cls(volume_size=vol, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
               embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
               decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
               decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
               norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
               norm_pix_loss=cfg.norm_pix_loss, args=args)
