"""AdamW over the engine's flat arenas (one fused streaming kernel per weight-decay segment).

``FusedAdamW(model, ...)`` is what new code should construct.  ``adopt(optimizer, model)`` lets an
unchanged reference script keep its ``torch.optim.AdamW(optim_factory.add_weight_decay(model, wd),
lr=..., betas=(0.9, 0.95))`` (k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:168-169):
the torch optimizer object stays the owner of the hyper-parameters and of a (view-backed)
``state_dict``, while ``step()`` runs ``vitae_adamw_step`` on the arena.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._abi import VitaeError


class FusedAdamW:
    """torch.optim-like facade: ``param_groups`` (lr is read from group 0 every step, so
    ``lr_sched.adjust_learning_rate`` works), ``zero_grad``, ``step``, ``state_dict``."""

    def __init__(self, model, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-8):
        self.model = model
        decay, no_decay = [], []
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if (p.ndim <= 1 or n.endswith('.bias')) else decay).append(p)
        self.param_groups = [{'params': no_decay, 'weight_decay': 0.0, 'lr': lr, 'betas': betas, 'eps': eps},
                             {'params': decay, 'weight_decay': weight_decay, 'lr': lr, 'betas': betas, 'eps': eps}]
        self.defaults = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        self._pending_state = None

    @property
    def engine(self):
        eng = self.model.engine
        if eng is None:
            return None
        if eng.opt_state is None:
            g = self.param_groups
            eng.init_optimizer(weight_decay=g[1]['weight_decay'], betas=g[1]['betas'], eps=g[1]['eps'])
            if self._pending_state is not None:
                self._restore(eng, self._pending_state)
                self._pending_state = None
        return eng

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    @torch.no_grad()
    def step(self):
        eng = self.engine
        if eng is None:
            raise VitaeError('FusedAdamW.step() before any forward/backward on the GPU')
        eng.weight_decay = self.param_groups[1]['weight_decay']
        eng.betas, eng.eps = self.param_groups[1]['betas'], self.param_groups[1]['eps']
        eng.optimizer_hparams(lr=self.param_groups[0]['lr'])
        eng.grad_norm_and_step()

    # ---- checkpointing: flat arenas + step (engine layout is deterministic for a given model)
    def state_dict(self):
        eng = self.engine
        state = {'step': 0, 'exp_avg': None, 'exp_avg_sq': None}
        if eng is not None:
            # (the wire format is fp32 whatever the engine stores: bf16 moments of the bf16 precision mode are widened here, narrowed by _restore)
            state = {'step': eng.read_opt_step(), 'exp_avg': eng.opt_state['exp_avg'].detach().float().cpu(),
                     'exp_avg_sq': eng.opt_state['exp_avg_sq'].detach().float().cpu(),
                     'layout': {k: (o, list(s)) for k, (o, s) in eng.layout.items()}}
        groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]
        return {'fused_adamw': state, 'param_groups': groups}

    def _restore(self, eng, state):
        if state.get('exp_avg') is not None:
            eng.opt_state['exp_avg'].copy_(state['exp_avg'])
            eng.opt_state['exp_avg_sq'].copy_(state['exp_avg_sq'])
        eng.write_opt_step(int(state.get('step', 0)))

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd.get('param_groups', [])):
            g.update(s)
        state = sd.get('fused_adamw')
        if state is None:
            raise VitaeError('not a FusedAdamW state dict')
        if self.model.engine is not None:
            self._restore(self.engine, state)
        else:
            self._pending_state = state


class _AdoptedAdamW:
    """Runs a foreign torch.optim.AdamW's step on the engine arenas (see module docstring)."""

    def __init__(self, optimizer: torch.optim.AdamW, model):
        self.optimizer, self.model = optimizer, model
        eng = model.engine
        ids_decay = {id(p) for n, p in model._trainable_named if not (p.ndim <= 1 or n.endswith('.bias'))}
        wd = None
        for g in optimizer.param_groups:
            if g.get('amsgrad') or g.get('maximize'):
                raise VitaeError('amsgrad / maximize are not supported by the fused AdamW')
            for p in g['params']:
                if (id(p) in ids_decay) != (g['weight_decay'] != 0.0) and g['weight_decay'] != 0.0:
                    raise VitaeError('optimizer groups do not follow the decay / no-decay split of the arena')
            if g['weight_decay'] != 0.0:
                wd = g['weight_decay']
        n_opt = sum(len(g['params']) for g in optimizer.param_groups)
        if n_opt != len(model._trainable):
            raise VitaeError('optimizer does not hold exactly the model parameters')
        g0 = optimizer.param_groups[0]
        eng.init_optimizer(weight_decay=wd or 0.0, betas=tuple(g0['betas']), eps=g0['eps'])
        # resume: import existing per-parameter state, then re-publish it as views of the arenas
        step = 0
        for n, p in model._trainable_named:
            st = optimizer.state.get(p)
            o, shp = eng.layout[n]
            k = p.numel()
            if st:
                eng.opt_state['exp_avg'][o:o + k].view(shp).copy_(st['exp_avg'])
                eng.opt_state['exp_avg_sq'][o:o + k].view(shp).copy_(st['exp_avg_sq'])
                step = max(step, int(st['step']))
            optimizer.state[p] = {'step': torch.tensor(float(step)),
                                  'exp_avg': eng.opt_state['exp_avg'][o:o + k].view(shp),
                                  'exp_avg_sq': eng.opt_state['exp_avg_sq'][o:o + k].view(shp)}
        eng.write_opt_step(step)
        self.engine = eng
        if not getattr(optimizer, '_vitae_hooked', False):
            optimizer.register_state_dict_pre_hook(lambda opt: opt._vitae_adopter._publish_step(opt))
            optimizer._vitae_hooked = True
        self._orig_step = optimizer.step
        optimizer.step = self.step
        optimizer.engine = eng

    def _publish_step(self, optimizer):
        step = self.engine.read_opt_step()
        for st in optimizer.state.values():
            st['step'] = torch.tensor(float(step))

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.optimizer.param_groups[0]
        eng = self.engine
        eng.betas, eng.eps = tuple(g0['betas']), g0['eps']
        eng.optimizer_hparams(lr=g0['lr'])
        eng.grad_norm_and_step()


def _adoptable_weight_decay(optimizer, model):
    """The single non-zero weight decay of a torch.optim.AdamW whose groups follow the arena's decay / no-decay split
    (timm add_weight_decay: no decay iff ndim <= 1 or name ends with '.bias'), or None when the optimiser cannot be
    re-routed without changing results."""
    ids_decay = {id(p) for n, p in model._trainable_named if not (p.ndim <= 1 or n.endswith('.bias'))}
    ids_all = {id(p) for p in model._trainable}
    wds, seen = set(), set()
    g0 = optimizer.param_groups[0]
    for g in optimizer.param_groups:
        if g.get('amsgrad') or g.get('maximize') or 'lr_scale' in g:
            return None
        if g['lr'] != g0['lr'] or tuple(g['betas']) != tuple(g0['betas']) or g['eps'] != g0['eps']:
            return None
        for p in g['params']:
            if id(p) not in ids_all or id(p) in seen:
                return None
            seen.add(id(p))
            if (id(p) in ids_decay) != (g['weight_decay'] != 0.0):
                return None         # e.g. a plain AdamW(model.parameters(), weight_decay=wd): biases in a decayed group
        if g['weight_decay'] != 0.0:
            wds.add(g['weight_decay'])
    if seen != ids_all or len(wds) > 1:
        return None
    return wds.pop() if wds else 0.0


def adopt(optimizer, model) -> Optional[object]:
    """Make ``optimizer.step()`` run on the HIP engine when that is possible without changing results:
    FusedAdamW is returned as is; a plain torch.optim.AdamW over exactly the model's parameters, grouped like
    timm's add_weight_decay, is re-routed in place (again, with its state migrated, when the model rebuilt its engine after
    ``set_precision()`` / ``.to()``); anything else is left alone (returns None: the caller keeps torch's own step)."""
    if isinstance(optimizer, FusedAdamW):
        return optimizer
    if model.engine is None:
        return None
    if getattr(optimizer, 'engine', None) is model.engine:
        return optimizer
    if type(optimizer) is torch.optim.AdamW and _adoptable_weight_decay(optimizer, model) is not None:
        if getattr(optimizer, 'engine', None) is not None:      # adopted for an engine that no longer exists
            # the per-parameter 'step' tensors are only refreshed by the state_dict pre-hook: publish the old engine's count
            # now, so that the new adopter resumes from it (else the bias corrections restart next to warmed-up moments)
            optimizer._vitae_adopter._publish_step(optimizer)
            optimizer.step = optimizer._vitae_adopter._orig_step
        optimizer._vitae_adopter = _AdoptedAdamW(optimizer, model)
        return optimizer
    return None
