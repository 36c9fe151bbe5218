"""CPU restatement of the reference's MAE(+contrastive) training loop on top of ``mae_ref``.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Restates
``utils/train_one_epoch.py:21-110`` (``train_one_stage_epoch``), ``utils/lr_sched.py:9-21`` and
the optimiser set-up of ``k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:157-171``
(timm ``add_weight_decay`` + ``torch.optim.AdamW(betas=(0.9, 0.95))``; the GradScaler of
``utils/misc.py:251-277`` is a no-op on a CPU host and a power-of-two scale in fp32 otherwise).
Also the timed CPU baseline of ``bench.py`` (``cpu_baseline.kind == "port"``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional

import torch

from . import mae_ref as R


def lr_at(epoch_float: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """utils/lr_sched.py:9-21."""
    if epoch_float < warmup_epochs:
        return lr * epoch_float / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (
        1. + math.cos(math.pi * (epoch_float - warmup_epochs) / (epochs - warmup_epochs)))


class RefTrainer:
    """Holds leaf params + AdamW + BatchNorm running state for the functional oracle."""

    def __init__(self, cfg: R.RefConfig, state_dict: Dict[str, torch.Tensor], lr: float = 1e-3,
                 weight_decay: float = 0.05, betas=(0.9, 0.95), eps: float = 1e-8,
                 dtype=torch.float32):
        self.cfg = cfg
        self.params = R.make_leaf_params(state_dict, dtype=dtype)
        groups = R.param_groups(self.params, weight_decay)
        self.optimizer = torch.optim.AdamW(
            [{'params': [self.params[n] for n in g['names']], 'weight_decay': g['weight_decay']}
             for g in groups], lr=lr, betas=betas, eps=eps)
        self.bn_state = None
        if cfg.contrastive:
            self.bn_state = {k: self.params['predictor.1.' + k]
                             for k in ('running_mean', 'running_var', 'num_batches_tracked')}

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def forward_losses(self, view1, view2, noise1, noise2, mask_ratio, edge_map_weight,
                       contr_weight):
        """One forward of the training objective: returns (total, terms dict, outputs)."""
        cfg = self.cfg
        if cfg.contrastive:
            loss, pred, mask, p1, p2, z1, z2 = R.contr_forward(
                self.params, view1, view2, noise1, noise2, cfg, mask_ratio, edge_map_weight,
                bn_state=self.bn_state, training=True)
            contr = R.contrastive_loss(p1, p2, z1, z2, contr_weight)
        else:
            loss, pred, mask = R.mae_forward(self.params, view1, noise1, cfg, mask_ratio,
                                             edge_map_weight)
            contr = torch.zeros((), dtype=pred.dtype)
            p1 = p2 = None
        total = loss[0] + contr
        terms = {'loss': total, 'edge_map_loss': loss[1], 'reconstruction_loss': loss[2],
                 'perceptual_loss': loss[3], 'contr_loss': contr}
        return total, terms, {'pred': pred, 'mask': mask, 'p1': p1, 'p2': p2}

    def grad_norm(self) -> torch.Tensor:
        """utils/misc.py:280-292 (norm of per-parameter L2 norms)."""
        gs = [p.grad for p in self.params.values() if p.requires_grad and p.grad is not None]
        return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in gs]), 2.0)

    def step(self, view1, view2, noise1, noise2, *, lr: float, mask_ratio=0.75,
             edge_map_weight=0.0, contr_weight=0.001, accum_iter: int = 1, update: bool = True):
        """utils/train_one_epoch.py:44-74 for one iteration."""
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        total, terms, outs = self.forward_losses(view1, view2, noise1, noise2, mask_ratio,
                                                 edge_map_weight, contr_weight)
        (total / accum_iter).backward()
        norm = None
        if update:
            norm = self.grad_norm()
            self.optimizer.step()
            self.optimizer.zero_grad()
        return {k: float(v.detach()) for k, v in terms.items()}, norm, outs


def train_one_stage_epoch_ref(trainer: RefTrainer, batches: Iterable, epoch: int, *, lr: float,
                              min_lr: float = 0.0, warmup_epochs: float = 40, epochs: float = 50,
                              mask_ratio: float = 0.75, contr_weight: float = 0.001,
                              edge_map_weight: float = 0.0, accum_iter: int = 1,
                              noises: Optional[List] = None) -> Dict[str, float]:
    """utils/train_one_epoch.py:21-110: returns the epoch means
    {lr, edge_map_loss, reconstruction_loss, perceptual_loss, contr_loss, loss}.
    ``batches`` yields (sample, original_volume, label); ``noises[i]`` = (noise1, noise2)."""
    batches = list(batches)
    n = len(batches)
    sums = {k: 0.0 for k in ('lr', 'edge_map_loss', 'reconstruction_loss', 'perceptual_loss',
                             'contr_loss', 'loss')}
    cur_lr = trainer.optimizer.param_groups[0]['lr']
    trainer.optimizer.zero_grad()
    for it, (sample, original, _) in enumerate(batches):
        if it % accum_iter == 0:
            cur_lr = lr_at(it / n + epoch, lr, min_lr, warmup_epochs, epochs)
        n1, n2 = noises[it]
        terms, _, _ = trainer.step(sample, original, n1, n2, lr=cur_lr, mask_ratio=mask_ratio,
                                   edge_map_weight=edge_map_weight, contr_weight=contr_weight,
                                   accum_iter=accum_iter, update=(it + 1) % accum_iter == 0)
        if not math.isfinite(terms['loss']):
            raise FloatingPointError(f"Loss is {terms['loss']}, stopping training")
        for k in terms:
            sums[k] += terms[k]
        sums['lr'] += cur_lr
    return {k: v / n for k, v in sums.items()}
