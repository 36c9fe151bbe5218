"""Make ``import model...`` / ``import utils...`` (the reference's top-level package names,
e.g. ``from model.model_factory import get_models``, ``from utils import misc, lr_sched`` in
k_fold_training_scripts/k_fold_cross_valid_combined_brats.py) resolve to this package.

    import vit_ae_plus_plus_amd.dropin; vit_ae_plus_plus_amd.dropin.install()
    # ... then run the reference's pre-training script unchanged
"""
import importlib
import sys

_ALIASES = {
    'model': 'vit_ae_plus_plus_amd.model',
    'model.model_factory': 'vit_ae_plus_plus_amd.model.model_factory',
    'model.vit': 'vit_ae_plus_plus_amd.model.vit',
    'model.vit_autoenc': 'vit_ae_plus_plus_amd.model.vit_autoenc',
    'model.model_utils': 'vit_ae_plus_plus_amd.model.model_utils',
    'model.model_utils.vit_helpers': 'vit_ae_plus_plus_amd.model.model_utils.vit_helpers',
    'model.model_utils.sobel_filter': 'vit_ae_plus_plus_amd.model.model_utils.sobel_filter',
    'model.model_utils.gaussian_filter': 'vit_ae_plus_plus_amd.model.model_utils.gaussian_filter',
    'model.model_utils.perceptual_loss': 'vit_ae_plus_plus_amd.model.model_utils.perceptual_loss',
    'utils': 'vit_ae_plus_plus_amd.utils',
    'utils.misc': 'vit_ae_plus_plus_amd.utils.misc',
    'utils.lr_sched': 'vit_ae_plus_plus_amd.utils.lr_sched',
    'utils.train_one_epoch': 'vit_ae_plus_plus_amd.utils.train_one_epoch',
    'utils.custom_loss': 'vit_ae_plus_plus_amd.utils.custom_loss',
}


def install(force: bool = False):
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not force and sys.modules[alias].__name__ != target:
            raise ImportError(f'{alias} is already imported from elsewhere ({sys.modules[alias]}); '
                              f'call install() before importing the reference packages or pass force=True')
        sys.modules[alias] = importlib.import_module(target)
