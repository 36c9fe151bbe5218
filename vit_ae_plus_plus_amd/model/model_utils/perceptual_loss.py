"""Perceptual-loss hook (reference: model/model_utils/perceptual_loss.py:11-77).

In the reference this term is computed under no_grad from ``torch.as_tensor`` values, so it never
contributes gradient, and its weight is 0 in the shipped configuration; it also needs
torchvision's VGG16 plus an unshipped checkpoint.  It is kept as a zero-valued logging hook.
"""
import torch
from torch import nn


class vgg_perceptual_loss(nn.Module):
    def __init__(self, requires_grad=False, use_imagenet=False):
        super().__init__()

    def forward(self, pred_vol, target_vol):
        return torch.zeros((), device=pred_vol.device)
