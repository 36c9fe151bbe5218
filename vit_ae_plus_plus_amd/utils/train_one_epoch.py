"""Per-epoch training loops with the reference's signatures
(reference: utils/train_one_epoch.py:21-110 ``train_one_stage_epoch``, :113-114
``compute_contrastive_loss``, :117-181 ``train_one_epoch``).

``train_one_stage_epoch`` keeps the reference's observable behaviour — same positional arguments,
same meters, same returned ``{name: epoch mean}`` dict, ``sys.exit(1)`` on a non-finite loss, the
same tensorboard tags — but when the model is a ``vit_ae_plus_plus_amd`` MAE and the optimiser is
(adoptable as) the fused AdamW it drives the whole optimisation step as one launch sequence of
``libvitae_hip.so`` kernels (optionally a captured HIP graph), reads the six loss scalars back with
one asynchronous copy per iteration, and never calls ``torch.cuda.synchronize`` /
``torch.cuda.empty_cache`` per step.  Anything else (foreign
optimiser, gradient clipping) takes the generic autograd route through the same kernels.
"""
from __future__ import annotations

import math
import sys
from collections import deque
from typing import Iterable

import torch

from .. import optim as fused_optim
from .._abi import CONSTS, lib
from ..engine import HP
from . import lr_sched, misc


# ----------------------------------------------------------------------------- contrastive loss
class _Scratch:
    _per_device = {}

    def __init__(self, device):
        self.hp = torch.zeros(CONSTS['VITAE_HP_COUNT'], dtype=torch.float32, device=device)
        self.acc = torch.zeros(CONSTS['VITAE_ACC_COUNT'], dtype=torch.float64, device=device)
        self.out = torch.zeros(1, dtype=torch.float32, device=device)

    @classmethod
    def get(cls, device):
        s = cls._per_device.get(device)
        if s is None:
            s = cls._per_device[device] = cls(device)
        return s


class _CosineLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, z1, z2, weight):
        s = _Scratch.get(p1.device)
        st = torch.cuda.current_stream(p1.device).cuda_stream
        p1, p2, z1, z2 = (t.contiguous() for t in (p1, p2, z1, z2))
        R, D = p1.shape
        s.hp[HP['CONTR_W']] = weight
        lib.vitae_memset_zero(s.acc.data_ptr(), s.acc.numel() * 8, st)
        lib.vitae_cosine_loss_fwd(p1.data_ptr(), z2.data_ptr(), p2.data_ptr(), z1.data_ptr(), s.acc.data_ptr(),
                                  s.hp.data_ptr(), s.out.data_ptr(), R, D, st)
        ctx.save_for_backward(p1, p2, z1, z2)
        ctx.weight = weight
        return s.out[0].clone()

    @staticmethod
    def backward(ctx, g):
        p1, p2, z1, z2 = ctx.saved_tensors
        s = _Scratch.get(p1.device)
        st = torch.cuda.current_stream(p1.device).cuda_stream
        R, D = p1.shape
        s.hp[HP['G_CONTR']] = g * ctx.weight
        dp1, dp2 = torch.empty_like(p1), torch.empty_like(p2)
        lib.vitae_cosine_loss_bwd(p1.data_ptr(), z2.data_ptr(), p2.data_ptr(), z1.data_ptr(), s.hp.data_ptr(),
                                  dp1.data_ptr(), dp2.data_ptr(), R, D, st)
        return dp1, dp2, None, None, None


def compute_contrastive_loss(args, criterion, p1, p2, z1, z2):
    """args.contr_weight * (-(cos(p1, z2).mean() + cos(p2, z1).mean()) / 2); ``criterion`` is the
    reference's nn.CosineSimilarity(dim=1) and is honoured for non-GPU tensors only by raising."""
    if not p1.is_cuda:
        raise RuntimeError('compute_contrastive_loss: MI355X only (no CPU fallback)')
    return _CosineLoss.apply(p1, p2, z1, z2, float(args.contr_weight))


# ----------------------------------------------------------------------------- fused loop
def _is_hip_mae(model) -> bool:
    from ..model.vit_autoenc import MaskedAutoencoderViT
    return isinstance(model, MaskedAutoencoderViT)


class _Readback:
    """Ring of pinned host buffers for the six loss scalars (one async D2H per iteration)."""

    def __init__(self, depth=4):
        self.slots = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.pending = deque()
        self.i = 0

    def push(self, losses_dev, meta):
        k = self.i % len(self.slots)
        if len(self.pending) == len(self.slots):
            raise RuntimeError('readback ring overrun')
        self.slots[k].copy_(losses_dev, non_blocking=True)
        self.events[k].record()
        self.pending.append((k, meta))
        self.i += 1

    def pop(self, block):
        if not self.pending:
            return None
        k, meta = self.pending[0]
        if not block and not self.events[k].query():
            return None
        self.events[k].synchronize()
        self.pending.popleft()
        return self.slots[k].tolist(), meta


def _fused_epoch(model, data_loader, optimizer, device, epoch, log_writer, args, edge_map_weight, metric_logger,
                 header, print_freq):
    eng = model._ensure_engine(device)
    accum_iter = args.accum_iter
    contr = eng.cfg.contrastive
    contr_w = float(getattr(args, 'contr_weight', 0.0)) if contr else 0.0
    world = model._reducer.world_size if model._reducer is not None else 1
    eng.set_loss_weights(edge_map_weight, contr_w, accum_iter, world)
    world = misc.get_world_size()        # metric collectives follow the process group, exchanged gradients or not
    use_graph = bool(getattr(args, 'hip_graph', True))
    rb = _Readback()
    n_iter = len(data_loader)

    rows = []      # (iteration, lr, [loss, recon, edge, percep, contr]) not yet written to the log writer

    def emit(block):
        """Tensorboard scalars of a window of iterations; under data parallelism the window is averaged over the ranks
        with ONE collective.  Every rank calls this at the same iterations (``flush`` below), so the metric collective
        sits at the same place in every rank's sequence of collectives — the reference's five blocking all-reduces per
        step (utils/train_one_epoch.py:83-88), issued here whenever a read-back happened to be ready, could interleave
        differently with the gradient buckets on different ranks."""
        red = misc.all_reduce_mean_rows([r[2] for r in block])
        if log_writer is None:
            return
        for (it, lr, _), v in zip(block, red):
            if (it + 1) % accum_iter == 0:
                x = int((it / n_iter + epoch) * 1000)
                log_writer.add_scalar('train_loss', v[0], x)
                log_writer.add_scalar('lr', lr, x)
                log_writer.add_scalar('reconstruction_loss', v[1], x)
                log_writer.add_scalar('sobel_loss', v[2], x)
                log_writer.add_scalar('perceptual_loss', v[3], x)
                log_writer.add_scalar('contr_loss', v[4], x)

    def consume(limit):
        """Process finished read-backs; block only while more than ``limit`` steps are in flight.  No collective here."""
        while rb.pending:
            got = rb.pop(block=len(rb.pending) > limit)
            if got is None:
                return
            vals, (it, lr, stepped) = got
            weighted, edge, recon, percep, contr_loss = vals[0], vals[1], vals[2], vals[3], vals[4] if contr else 0.0
            loss_value = weighted + contr_loss
            metric_logger.update(edge_map_loss=edge, reconstruction_loss=recon, perceptual_loss=percep,
                                 contr_loss=contr_loss)
            if not math.isfinite(loss_value):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            # (a step the AdamW kernels skipped for a non-finite gradient norm — GradScaler.step semantics — is not counted:
            # the count of applied steps lives on the device, hp[VITAE_HP_STEP], and the bias corrections follow it)
            metric_logger.update(loss=loss_value)
            metric_logger.update(lr=lr)
            rows.append((it, lr, [loss_value, recon, edge, percep, contr_loss]))
            if world == 1:
                emit(rows)
                rows.clear()

    def flush():
        """Deterministic point (same iteration on every rank): drain the read-backs, then one metric collective."""
        consume(limit=0)
        if world > 1:
            emit(rows)
            rows.clear()

    for it, (sample, original_volume, _) in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        if it % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        lr = optimizer.param_groups[0]["lr"]
        update = (it + 1) % accum_iter == 0
        accumulate = it % accum_iter != 0
        runner = model._step_runner(sample.shape[0], float(args.mask_ratio), update, accumulate, use_graph)
        runner.load(sample, original_volume if contr else None)
        if update:
            optimizer_hparams(optimizer, eng, lr)
        runner.run()
        if update:
            for p in model._trainable:   # mirror optimizer.zero_grad(): grads are consumed
                p.grad = None
        rb.push(eng.losses, (it, lr, update))
        if world > 1 and (it + 1) % print_freq == 0:
            flush()
        else:
            consume(limit=2)
    flush()


def optimizer_hparams(optimizer, eng, lr):
    g = optimizer.param_groups
    decayed = [x for x in g if x.get('weight_decay', 0.0) != 0.0]
    eng.weight_decay = decayed[0]['weight_decay'] if decayed else 0.0
    eng.betas, eng.eps = tuple(g[0]['betas']), g[0]['eps']
    eng.optimizer_hparams(lr=lr)


# ----------------------------------------------------------------------------- public loops
def train_one_stage_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer,
                          device: torch.device, epoch: int, loss_scaler, log_writer=None, args=None,
                          edge_map_weight=0):
    model.train(True)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', misc.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    print_freq = 20
    accum_iter = args.accum_iter
    device = torch.device(device)
    optimizer.zero_grad()
    if log_writer is not None:
        print('log_dir: {}'.format(log_writer.log_dir))

    fused = False
    # (a perceptual weight keeps the generic route: the VGG hook is evaluated outside the captured step)
    if (_is_hip_mae(model) and device.type == 'cuda' and not getattr(args, 'no_fused_step', False)
            and not getattr(model, 'perceptual_weight', 0)):
        model._ensure_engine(device)
        fused = fused_optim.adopt(optimizer, model) is not None
    if fused:
        _fused_epoch(model, data_loader, optimizer, device, epoch, log_writer, args, edge_map_weight, metric_logger,
                     header, print_freq)
    else:
        criterion = torch.nn.CosineSimilarity(dim=1)
        n_iter = len(data_loader)
        red = getattr(model, '_reducer', None)
        if red is not None and red.active:
            # generic route under data parallelism: the exchange the fused step does inside its launch list happens here,
            # between backward and the optimiser step (every bucket, blocking, then the 1/world of the mean)
            from .. import ddp
            if not hasattr(loss_scaler, 'grad_sync'):
                raise RuntimeError('data parallel without the fused step needs utils.misc.NativeScalerWithGradNormCount '
                                   '(its grad_sync hook all-reduces the gradients before optimizer.step())')
            loss_scaler.grad_sync = lambda: ddp.allreduce_mean_now(red)
        rows = []
        for it, (sample, original_volume, _) in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
            if it % accum_iter == 0:
                lr_sched.adjust_learning_rate(optimizer, it / n_iter + epoch, args)
            sample = sample.to(device, non_blocking=True)
            original_volume = original_volume.to(device, non_blocking=True)
            loss, pred, mask, p1, p2, z1, z2 = model(view1=sample, view2=original_volume, mask_ratio=args.mask_ratio,
                                                     edge_map_weight=edge_map_weight)
            contr_loss = compute_contrastive_loss(args, criterion, p1, p2, z1, z2)
            edge_map_loss, reconstruction_loss, perceptual_loss = loss[1], loss[2], loss[3]
            total = loss[0] + contr_loss
            vals = torch.stack([total.detach(), edge_map_loss.detach(), reconstruction_loss.detach(),
                                perceptual_loss.detach(), contr_loss.detach()]).tolist()   # one sync
            loss_value = vals[0]
            metric_logger.update(edge_map_loss=vals[1], reconstruction_loss=vals[2], perceptual_loss=vals[3],
                                 contr_loss=vals[4])
            if not math.isfinite(loss_value):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            total = total / accum_iter
            loss_scaler(total, optimizer, parameters=model.parameters(), update_grad=(it + 1) % accum_iter == 0)
            if (it + 1) % accum_iter == 0:
                optimizer.zero_grad()
            metric_logger.update(loss=loss_value)
            lr = optimizer.param_groups[0]["lr"]
            metric_logger.update(lr=lr)
            rows.append((it, lr, [loss_value, vals[2], vals[1], vals[3], vals[4]]))
            if misc.get_world_size() == 1 or (it + 1) % print_freq == 0 or it + 1 == n_iter:
                # one metric collective per logging window, at the same iteration on every rank
                block = misc.all_reduce_mean_rows([r[2] for r in rows])
                if log_writer is not None:
                    for (i2, lr2, _), v in zip(rows, block):
                        if (i2 + 1) % accum_iter == 0:
                            x = int((i2 / n_iter + epoch) * 1000)
                            log_writer.add_scalar('train_loss', v[0], x)
                            log_writer.add_scalar('lr', lr2, x)
                            log_writer.add_scalar('reconstruction_loss', v[1], x)
                            log_writer.add_scalar('sobel_loss', v[2], x)
                            log_writer.add_scalar('perceptual_loss', v[3], x)
                            log_writer.add_scalar('contr_loss', v[4], x)
                rows.clear()
        if hasattr(loss_scaler, 'grad_sync'):
            loss_scaler.grad_sync = None

    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler,
                    max_norm=0, log_writer=None, args=None):
    """Pure-contrastive loop over a ``VisionTransformer3DContrastive`` (reference
    utils/train_one_epoch.py:117-181): model(original, augmented) -> (p1, p2, z1, z2)."""
    model.train(True)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', misc.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    accum_iter = args.accum_iter
    optimizer.zero_grad()
    if log_writer is not None:
        print('log_dir: {}'.format(log_writer.log_dir))
    n_iter = len(data_loader)
    for it, (augmented, original, _) in enumerate(metric_logger.log_every(data_loader, 20, header)):
        if it % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        augmented = augmented.to(device, non_blocking=True)
        original = original.to(device, non_blocking=True)
        p1, p2, z1, z2 = model(original, augmented)
        loss = -(criterion(p1, z2).mean() + criterion(p2, z1).mean()) * 0.5
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        loss = loss / accum_iter
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False,
                    update_grad=(it + 1) % accum_iter == 0)
        if (it + 1) % accum_iter == 0:
            optimizer.zero_grad()
        metric_logger.update(loss=loss_value)
        lrs = [g["lr"] for g in optimizer.param_groups]
        metric_logger.update(lr=max(lrs))
        loss_value_reduce = misc.all_reduce_mean(loss_value)
        if log_writer is not None and (it + 1) % accum_iter == 0:
            x = int((it / n_iter + epoch) * 1000)
            log_writer.add_scalar('loss', loss_value_reduce, x)
            log_writer.add_scalar('lr', max(lrs), x)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
