"""Per-iteration learning-rate schedule (reference: utils/lr_sched.py:9-21)."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    """Linear warm-up to ``args.lr`` over ``args.warmup_epochs`` (fractional epochs), then a
    half-cosine down to ``args.min_lr`` at ``args.epochs``; groups may carry an ``lr_scale``."""
    warm, total = args.warmup_epochs, args.epochs
    if epoch < warm:
        lr = args.lr * epoch / warm
    else:
        phase = math.pi * (epoch - warm) / (total - warm)
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1. + math.cos(phase))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
