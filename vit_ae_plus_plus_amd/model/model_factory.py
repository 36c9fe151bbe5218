"""String -> model constructor (reference: model/model_factory.py:9-29)."""
from functools import partial

from torch import nn

from . import vit_autoenc
from .vit import VisionTransformer3D, VisionTransformer3DContrastive


def get_models(model_name, args):
    if model_name in ('autoenc', 'autoenc_contr'):
        print(f"Number of channels is {args.in_channels}")
        ctor = vit_autoenc.__dict__[args.model]
        return ctor(volume_size=args.volume_size, in_chans=args.in_channels, patch_size=args.patch_size, args=args)
    if model_name == 'vit':
        return VisionTransformer3D(volume_size=args.volume_size, in_chans=args.in_channels,
                                   num_classes=args.nb_classes, patch_size=args.patch_size,
                                   global_pool=args.global_pool, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                   drop_path_rate=args.drop_path)
    if model_name == 'contrastive':
        return VisionTransformer3DContrastive(volume_size=args.volume_size, in_chans=args.in_channels,
                                              num_classes=args.nb_classes, patch_size=args.patch_size,
                                              global_pool=args.global_pool,
                                              norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                              drop_path_rate=args.drop_path, use_proj=args.use_proj)
    raise NotImplementedError("Only AE model supported till now")
