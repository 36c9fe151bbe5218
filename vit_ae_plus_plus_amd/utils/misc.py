"""Training-loop utilities with the reference's names (reference: utils/misc.py:24-340):
meters / logger, distributed init, loss scaler + gradient norm, checkpoint save / load,
``all_reduce_mean``.

Differences that matter on MI355X (all value-preserving):
  * meters accept already-synchronised python floats, so the fused loop reads all loss scalars back
    with ONE device->host copy per iteration instead of five ``.item()`` syncs
    (reference utils/train_one_epoch.py:59-64);
  * ``all_reduce_means`` reduces a list of scalars in one RCCL call (reference: five blocking
    single-scalar all-reduces per step, utils/train_one_epoch.py:83-88);
  * fp32 gradients need no loss scaling (bf16 has fp32's exponent range), so
    ``NativeScalerWithGradNormCount`` keeps the GradScaler *state-dict format* for checkpoint
    compatibility but applies scale 1.0; the inf/nan guard of ``GradScaler.step`` is done by the fused
    AdamW kernel, which skips the update when the global gradient norm is not finite.
"""
from __future__ import annotations

import builtins
import datetime
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import torch
import torch.distributed as dist

inf = float('inf')


# ----------------------------------------------------------------------------- meters
class SmoothedValue:
    """Windowed (median / mean / max / last) and global (mean) statistics of a scalar series."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = fmt if fmt is not None else "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """Sums ``count`` and ``total`` over ranks (the window is left local, as in the reference)."""
        if not is_dist_avail_and_initialized():
            return
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        c, s = t.tolist()
        self.count, self.total = int(c), s

    @property
    def median(self):
        vals = sorted(self.deque)   # torch.median returns the lower middle element
        return float(vals[(len(vals) - 1) // 2]) if vals else float('nan')

    @property
    def avg(self):
        return float(torch.tensor(list(self.deque), dtype=torch.float32).mean()) if self.deque else float('nan')

    @property
    def global_avg(self):
        return self.total / self.count if self.count else float('nan')

    @property
    def max(self):
        return max(self.deque) if self.deque else float('nan')

    @property
    def value(self):
        return self.deque[-1] if self.deque else float('nan')

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for name, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[name].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(k, str(m)) for k, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ''
        n = len(iterable)
        width = len(str(n))
        iter_time, data_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        on_gpu = torch.cuda.is_available()
        t_start = t_prev = time.time()
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - t_prev)
            yield obj
            iter_time.update(time.time() - t_prev)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                fields = [header, f'[{i:{width}d}/{n}]', f'eta: {eta}', str(self), f'time: {iter_time}',
                          f'data: {data_time}']
                if on_gpu:
                    fields.append('max mem: {:.0f}'.format(torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)))
                print(self.delimiter.join(fields))
            t_prev = time.time()
        total = time.time() - t_start
        print('{} Total time: {} ({:.4f} s / it)'.format(header, str(datetime.timedelta(seconds=int(total))),
                                                         total / max(n, 1)))


# ----------------------------------------------------------------------------- distributed
def setup_for_distributed(is_master):
    """Silences ``print`` on non-master ranks (``force=True`` overrides)."""
    raw = getattr(builtins, '_vitae_raw_print', builtins.print)
    builtins._vitae_raw_print = raw

    def quiet_print(*args, **kwargs):
        force = kwargs.pop('force', False) or get_world_size() > 8
        if is_master or force:
            raw('[{}] '.format(datetime.datetime.now().time()), end='')
            raw(*args, **kwargs)

    builtins.print = quiet_print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """One process per GPU over RCCL (``backend='nccl'`` is RCCL on ROCm).  Rank discovery as in the
    reference (utils/misc.py:216-248): OpenMPI, torchrun env, SLURM; otherwise single process.
    Unlike the reference this does not require ``args.dist_url`` to exist: it defaults
    to ``env://``."""
    env = os.environ
    if getattr(args, 'dist_on_itp', False):
        args.rank, args.world_size = int(env['OMPI_COMM_WORLD_RANK']), int(env['OMPI_COMM_WORLD_SIZE'])
        args.gpu = int(env['OMPI_COMM_WORLD_LOCAL_RANK'])
        args.dist_url = "tcp://%s:%s" % (env['MASTER_ADDR'], env['MASTER_PORT'])
        env['LOCAL_RANK'], env['RANK'], env['WORLD_SIZE'] = str(args.gpu), str(args.rank), str(args.world_size)
    elif 'RANK' in env and 'WORLD_SIZE' in env:
        args.rank, args.world_size, args.gpu = int(env['RANK']), int(env['WORLD_SIZE']), int(env['LOCAL_RANK'])
    elif 'SLURM_PROCID' in env:
        args.rank = int(env['SLURM_PROCID'])
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
        args.world_size = int(env.get('SLURM_NTASKS', getattr(args, 'world_size', 1)))
    else:
        print('Not using distributed mode')
        setup_for_distributed(is_master=True)
        args.distributed = False
        return
    args.distributed = True
    if not getattr(args, 'dist_url', None):
        args.dist_url = 'env://'
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = 'nccl' if use_gpu else 'gloo'
    print('| distributed init (rank {}): {}, gpu {}'.format(args.rank, args.dist_url, args.gpu), flush=True)
    kw = dict(device_id=torch.device('cuda', args.gpu)) if use_gpu else {}
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank, **kw)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def all_reduce_means(values):
    """Mean over ranks of a list of python scalars with ONE collective."""
    ws = get_world_size()
    if ws == 1:
        return list(values)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    return (t / ws).tolist()


def all_reduce_mean(x):
    if get_world_size() == 1:
        return x
    return all_reduce_means([float(x)])[0]


def all_reduce_mean_rows(rows):
    """Mean over ranks of a list of equally long rows of python scalars with ONE collective (the per-iteration metric
    rows of a logging window, utils/train_one_epoch.py)."""
    if get_world_size() == 1 or not rows:
        return [list(r) for r in rows]
    n = len(rows[0])
    flat = all_reduce_means([v for r in rows for v in r])
    return [flat[i * n:(i + 1) * n] for i in range(len(rows))]


# ----------------------------------------------------------------------------- scaler / grad norm
def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """Global gradient norm (reference utils/misc.py:280-292).  For parameters living in a
    HipMAEEngine arena use ``engine_grad_norm`` (one streaming kernel); this generic version is the
    host-driven equivalent for foreign parameters."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    if float(norm_type) == inf:
        return max(g.abs().max() for g in grads)
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, norm_type) for g in grads]), norm_type)


def engine_grad_norm(engine) -> torch.Tensor:
    """L2 norm of the whole gradient arena by ``vitae_grad_sqnorm`` -> 0-dim device tensor."""
    from .._abi import lib
    st = torch.cuda.current_stream(engine.device).cuda_stream
    lib.vitae_memset_zero(engine.acc.data_ptr(), engine.acc.numel() * 8, st)   # scalars were already read out
    lib.vitae_grad_sqnorm(engine.grads.data_ptr(), engine.n_total, engine.acc.data_ptr(),
                          engine.losses.data_ptr() + 20, st)
    return engine.losses[5]


class NativeScalerWithGradNormCount:
    """``loss_scaler(loss, optimizer, clip_grad=None, parameters=..., update_grad=True)``
    (reference utils/misc.py:251-277): backward, global grad norm, optimizer step.  Scale is fixed
    at 1.0 (see module docstring); the state dict keeps torch.cuda.amp.GradScaler's keys."""
    state_dict_key = "amp_scaler"

    def __init__(self):
        self._state = {'scale': 1.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000,
                       '_growth_tracker': 0}
        # data parallel, generic (non-fused) route: called after backward and before the norm / the step so that every
        # rank steps on the mean gradient (set by train_one_stage_epoch; the fused step exchanges inside its launch list)
        self.grad_sync = None

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        params = list(parameters) if parameters is not None else [p for g in optimizer.param_groups for p in g['params']]
        engine = getattr(optimizer, 'engine', None)
        if self.grad_sync is not None:
            self.grad_sync()
        if clip_grad is not None:
            norm = torch.nn.utils.clip_grad_norm_(params, clip_grad)
        elif engine is not None:
            norm = None   # computed inside the fused step (losses[5])
        else:
            norm = get_grad_norm_(params)
        optimizer.step()
        if engine is not None and norm is None:
            norm = engine.losses[5]
        return norm

    def state_dict(self):
        return dict(self._state)

    def load_state_dict(self, state_dict):
        self._state.update({k: v for k, v in state_dict.items() if k in self._state and k != 'scale'})


# ----------------------------------------------------------------------------- checkpoints
def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler):
    """``checkpoint-<epoch>.pth`` = {'model','optimizer','epoch','scaler','args'} on rank 0
    (reference utils/misc.py:295-309)."""
    path = Path(args.output_dir) / ('checkpoint-%s.pth' % str(epoch))
    payload = {'model': model_without_ddp.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch,
               'args': args}
    if loss_scaler is not None:
        payload['scaler'] = loss_scaler.state_dict()
    save_on_master(payload, path)


def load_model(args, model_without_ddp, optimizer, loss_scaler):
    """Resume from ``args.resume`` (path or https URL): model, then optimizer + scaler when present
    and not in eval mode (reference utils/misc.py:315-329; start_epoch is not restored there either)."""
    if not getattr(args, 'resume', None):
        return
    if args.resume.startswith('https'):
        ckpt = torch.hub.load_state_dict_from_url(args.resume, map_location='cpu', check_hash=True)
    else:
        ckpt = torch.load(args.resume, map_location='cpu', weights_only=False)
    model_without_ddp.load_state_dict(ckpt['model'])
    print("Resume checkpoint %s" % args.resume)
    if 'optimizer' in ckpt and 'epoch' in ckpt and not getattr(args, 'eval', False):
        optimizer.load_state_dict(ckpt['optimizer'])
        if 'scaler' in ckpt and loss_scaler is not None:
            loss_scaler.load_state_dict(ckpt['scaler'])
        print("With optim & sched!")
