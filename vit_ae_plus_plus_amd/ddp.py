"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL / xGMI.

The reference's pre-training scripts never wrap the model in DDP (only scalar metric
all-reduces exist on that path, utils/misc.py:332-340), so this is new functionality with the
standard DDP contract: after the exchange every rank holds the MEAN of the per-rank gradients.

Design for MI355X: the gradient arena of ``HipMAEEngine`` is laid out in forward order, and the
hand-ordered backward finishes it in contiguous ranges (decoder+predictor, then the encoder in
``engine.enc_chunks`` groups of blocks from the top, the last with the patch embedding).  Each range is
one bucket (50-130 MB — large messages, since xGMI rings are per-link bound): its all-reduce is issued
asynchronously the moment the backward phase that completes it has been enqueued, so RCCL runs on its
own stream underneath the remaining backward kernels; the small token/vector segment goes last.
With 531 MB of fp32 gradients and ~3 ms of backward to hide them in, the exchange can optionally run
in bf16 (``comm_dtype``; the same trade as torch's ``bf16_compress_hook``: gradients are rounded to
bf16 for the wire and the sum, the optimiser still sees fp32).  The 1/world_size of the
mean is folded into the loss-gradient multipliers up front (``HipMAEEngine.set_loss_weights``), so
a SUM all-reduce yields the mean with no extra pass over the 0.5 GB arena.
The reducer only needs a flat tensor and ranges, so it is exercised on CPU with gloo in tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _ModelledRing:
    """DESIGN AID, never a measurement: on a world of ONE rank (``force``) the all-reduce costs nothing, so the shape of the
    exchange (how many buckets, which one is exposed behind the backward, which hardware queue the collective sits on) cannot
    be tuned on a one-GPU box.  With ``VITAE_DDP_SIM_BUSBW=<GB/s>`` every bucket additionally occupies the communication
    stream for the time a ring all-reduce over ``VITAE_DDP_SIM_WORLD`` (default 8) ranks would take at that bus bandwidth,
    2 (N-1)/N * bytes / busbw + a fixed latency.  It models the exposure only — not the CUs and HBM bandwidth RCCL's kernels
    take from the step."""

    def __init__(self, device):
        self.busbw = float(os.environ['VITAE_DDP_SIM_BUSBW']) * 1e9
        self.world = int(os.environ.get('VITAE_DDP_SIM_WORLD', '8'))
        self.latency = float(os.environ.get('VITAE_DDP_SIM_LATENCY_US', '30')) * 1e-6
        self.cycles_per_s = _spin_rate(device)

    def occupy(self, nbytes: int):
        """Called with the communication stream current."""
        t = 2.0 * (self.world - 1) / self.world * nbytes / self.busbw + self.latency
        torch.cuda._sleep(int(t * self.cycles_per_s))


def _spin_rate(device) -> float:
    """torch.cuda._sleep cycles per second."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    with torch.cuda.device(device):
        a.record()
        torch.cuda._sleep(10_000_000)
        b.record()
    torch.cuda.synchronize(device)
    return 10_000_000 / (a.elapsed_time(b) * 1e-3)


class _StreamWork:
    """Completion of a collective that ran on a stream of ours: waiting = the current stream OF THE BUCKET'S DEVICE waits for
    its event (the process's current device need not be the one the gradients live on)."""

    def __init__(self, ev, device=None):
        self.ev, self.device = ev, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.ev)


def held_back_by(busy: "torch.cuda.Stream", candidates, spin_ms: float = 4.0, rate: Optional[float] = None):
    """[bool per candidate]: does a kernel on the candidate stream wait for a long kernel on ``busy``?  HIP multiplexes its
    streams over GPU_MAX_HW_QUEUES (4) hardware queues; two streams on one queue run strictly one after the other, so a
    0.4 ms all-reduce kernel holds back every launch of a stream that shares its queue (tools/probes/hwq_probe.py: classes of
    three streams each on this runtime; raising GPU_MAX_HW_QUEUES makes the whole step 3x slower).  Measured, not derived: a
    spin on ``busy``, a tiny kernel + event on every candidate, event times compared."""
    dev = busy.device
    rate = rate or _spin_rate(dev)
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize(dev)
    t0, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in candidates]
    with torch.cuda.stream(busy):
        t0.record()
        torch.cuda._sleep(int(spin_ms * 1e-3 * rate))
        done.record()
    for s, ev in zip(candidates, evs):
        if s is busy or s.cuda_stream == busy.cuda_stream:
            continue
        with torch.cuda.stream(s):
            x.add_(1.0)
            ev.record()
    torch.cuda.synchronize(dev)
    total = t0.elapsed_time(done)
    return [True if (s is busy or s.cuda_stream == busy.cuda_stream) else t0.elapsed_time(ev) > 0.5 * total
            for s, ev in zip(candidates, evs)]


def pick_streams(device, main: Optional["torch.cuda.Stream"] = None, n_candidates: int = 8):
    """(optimiser stream, communication stream, report): two streams that share a hardware queue neither with ``main`` (the
    stream the step is launched on; default: the current one) nor with each other — so that the gradient all-reduce, the
    per-bucket AdamW and the backward really run side by side.  Falls back to fresh streams if no such pair is found."""
    device = torch.device(device)
    main = main or torch.cuda.current_stream(device)
    cand = [torch.cuda.Stream(device=device) for _ in range(n_candidates)]
    x = torch.zeros(64, device=device)
    for s in cand:                         # bind every candidate to its hardware queue (first use)
        with torch.cuda.stream(s):
            x.add_(1.0)
    rate = _spin_rate(device)
    with_main = held_back_by(main, cand, rate=rate)
    free = [s for s, hb in zip(cand, with_main) if not hb]
    if not free:
        return cand[0], cand[1], {'ok': False, 'with_main': with_main}
    opt = free[0]
    with_opt = held_back_by(opt, free, rate=rate)
    rest = [s for s, hb in zip(free, with_opt) if not hb]
    if not rest:
        return opt, free[-1], {'ok': False, 'with_main': with_main, 'with_opt': with_opt}
    return opt, rest[0], {'ok': True, 'with_main': with_main, 'with_opt': with_opt}


class GradBucketReducer:
    """Asynchronous SUM all-reduce of contiguous ranges of one flat gradient tensor."""

    def __init__(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None,
                 max_bucket_elems: Optional[int] = None, force: bool = False, comm_dtype: Optional[torch.dtype] = None,
                 cast_ranges: Optional[Sequence[Tuple[int, int]]] = None):
        assert flat.dim() == 1
        self.flat, self.group = flat, group
        self.comm_dtype = comm_dtype if comm_dtype not in (None, flat.dtype) else None
        self.wire = torch.zeros_like(flat, dtype=self.comm_dtype) if self.comm_dtype is not None else None
        self.force = force   # exercise the exchange machinery even at world size 1 (single-GPU validation)
        self.ranges: List[List[Tuple[int, int]]] = []
        for s, e in ranges:
            assert 0 <= s <= e <= flat.numel()
            parts = []
            if max_bucket_elems and e - s > max_bucket_elems:
                k = s
                while k < e:
                    parts.append((k, min(e, k + max_bucket_elems)))
                    k += max_bucket_elems
            else:
                parts.append((s, e))
            self.ranges.append(parts)
        # with a wire dtype: the sub-ranges that still need rounding into the wire buffer at launch time (None = all;
        # the engine's wgrad epilogues write the wire copy of the big matrices themselves)
        self.cast_ranges = None if cast_ranges is None else sorted(cast_ranges)
        self.pending = []
        self._model = _ModelledRing(flat.device) if (force and flat.is_cuda and os.environ.get('VITAE_DDP_SIM_BUSBW')
                                                     and not is_distributed()) else None
        # CUDA tensors: the collectives run on THIS stream (a synchronous torch.distributed call runs on the caller's current
        # stream — torch >= 2.7, checked on this image by tools/probes/hwq_probe2.py — so the stream, and with it the hardware
        # queue the all-reduce kernel occupies, is ours to choose: ``pick_streams``); None = asynchronous Work objects
        # (CPU / gloo, where there is no stream)
        self.comm_stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None
        # diagnostics (bench.py --gpus N): ``timing`` = a list collects (bucket, start event, end event) around every all-reduce on
        # the communication stream; ``paused`` switches the exchange off (replicas drift apart: measurement only)
        self.timing: Optional[list] = None
        self.paused = False

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if is_distributed() else 1

    @property
    def active(self) -> bool:
        if self.paused:
            return False
        return self.world_size > 1 or (self.force and dist.is_available() and dist.is_initialized())

    def timing_report(self):
        """-> {bucket: mean milliseconds of its all-reduce on the communication stream} from the events collected so far."""
        out = {}
        for b, e0, e1 in self.timing or []:
            out.setdefault(b, []).append(e0.elapsed_time(e1))
        return {b: sum(v) / len(v) for b, v in sorted(out.items())}

    def launch(self, bucket: int):
        """Issue the all-reduce of bucket ``bucket`` (non-blocking; ordered after work already enqueued
        on the current stream)."""
        if not self.active:
            return
        for s, e in self.ranges[bucket]:
            if e > s:
                if self.wire is not None:
                    # round to the wire dtype on the CURRENT stream: moving this pass to its own stream (to hide it under
                    # the next phase) made the step 0.35 ms slower — event hops between graph replays cost more
                    if self.cast_ranges is None:
                        self.wire[s:e].copy_(self.flat[s:e])
                    else:
                        for cs, ce in self.cast_ranges:
                            a, b = max(cs, s), min(ce, e)
                            if b > a:
                                self.wire[a:b].copy_(self.flat[a:b])
                    work = self._all_reduce(self.wire[s:e], bucket)
                else:
                    work = self._all_reduce(self.flat[s:e], bucket)
                self.pending.append((work, s, e, bucket))

    def _all_reduce(self, t: torch.Tensor, bucket: int = -1):
        if self.comm_stream is None:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        cs = self.comm_stream
        cs.wait_stream(torch.cuda.current_stream(t.device))      # ordered after the kernels that produced the bucket
        with torch.cuda.stream(cs):
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            if self.timing is not None:
                e1.record()
                self.timing.append((bucket, e0, e1))
            if self._model is not None:
                self._model.occupy(t.numel() * t.element_size())
            ev = torch.cuda.Event()
            ev.record()
        return _StreamWork(ev, self.flat.device)

    def wait(self, copy_back: bool = True):
        """Make the current stream (or the host, for CPU backends) wait for every launched bucket.  With a wire dtype,
        ``copy_back=False`` leaves the reduced values in ``self.wire`` only (the fused optimiser reads them there);
        the fp32 arena then still holds this rank's local gradients."""
        for w, s, e, _ in self.pending:
            w.wait()
            if self.wire is not None and copy_back:
                self.flat[s:e].copy_(self.wire[s:e])
        self.pending.clear()

    def wait_bucket(self, bucket: int, copy_back: bool = True):
        """Same for the pieces of one bucket only (the caller's current stream waits; the others stay pending)."""
        rest = []
        for w, s, e, b in self.pending:
            if b != bucket:
                rest.append((w, s, e, b))
                continue
            w.wait()
            if self.wire is not None and copy_back:
                self.flat[s:e].copy_(self.wire[s:e])
        self.pending = rest


class RcclBucketReducer(GradBucketReducer):
    """The same contract through the C ABI (``vitae_ddp_*``, csrc/ddp.hip): RCCL called directly on a side HIP stream that is
    forked from / joined to the compute stream with events, so the collectives are graph nodes of the captured step (one
    replay per optimisation step) instead of host-issued work between five graph replays.  ``torch.distributed`` is used for
    the rendezvous only (the 128-byte RCCL id travels by ``broadcast_object_list``)."""
    native = True

    _live = None         # weak reference to the reducer that owns the process's ONE communicator (csrc/ddp.hip keeps a single one)

    def __init__(self, flat, ranges, device, comm_dtype=None, cast_ranges=None, force=False, max_bucket_elems=None, group=None):
        from ._abi import VitaeError, lib
        import ctypes
        import weakref
        # the C ABI knows two wire formats (vitae_ddp_allreduce_bucket: fp32 or bf16); anything else would be summed with the wrong type
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise VitaeError(f'the native gradient exchange sends fp32 or bf16, not {comm_dtype}')
        if comm_dtype is torch.bfloat16 and flat.dtype is not torch.float32:
            raise VitaeError('bf16 wire format needs an fp32 gradient arena')
        # the communicator spans the WORLD (rendezvous below): a sub-group would broadcast parameters inside the group and
        # reduce gradients over everybody
        if group is not None and dist.is_available() and dist.is_initialized() and group is not dist.group.WORLD:
            raise VitaeError('the native gradient exchange runs over the default (world) process group only')
        prev = RcclBucketReducer._live() if RcclBucketReducer._live is not None else None
        if prev is not None and prev is not self and not getattr(prev, '_closed', False):
            raise VitaeError('a native gradient reducer (and step graphs captured with its communicator) is still alive: '
                             'close() it (model.disable_data_parallel()) before creating another one')
        super().__init__(flat, ranges, max_bucket_elems=max_bucket_elems, force=force, comm_dtype=comm_dtype, cast_ranges=cast_ranges)
        self._lib, self.device = lib, torch.device(device)
        self._closed = False
        if not lib.vitae_ddp_available():
            raise VitaeError('RCCL could not be bound (librccl.so): the native gradient exchange is unavailable')
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            lib.vitae_ddp_unique_id(ctypes.addressof(buf))
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0)
            buf = ctypes.create_string_buffer(box[0], 128)
        with torch.cuda.device(self.device):
            lib.vitae_ddp_init(ctypes.addressof(buf), world, rank)
        self._world = world
        RcclBucketReducer._live = weakref.ref(self)

    def close(self):
        """Give the process's communicator back (graphs captured with this reducer must not be replayed afterwards)."""
        if not self._closed:
            self._closed = True
            self.pending.clear()
            self._lib.vitae_ddp_destroy()

    @property
    def world_size(self) -> int:
        return self._world

    @property
    def active(self) -> bool:
        return (self._world > 1 or self.force) and not self.paused

    def launch(self, bucket: int):
        if not self.active:
            return
        cs = torch.cuda.current_stream(self.device).cuda_stream
        for s, e in self.ranges[bucket]:
            if e <= s:
                continue
            if self.wire is not None:
                if self.cast_ranges is None:
                    self.wire[s:e].copy_(self.flat[s:e])
                else:
                    for c0, c1 in self.cast_ranges:
                        a, b = max(c0, s), min(c1, e)
                        if b > a:
                            self.wire[a:b].copy_(self.flat[a:b])
                self._lib.vitae_ddp_allreduce_bucket(self.wire.data_ptr() + 2 * s, e - s, 1, cs, self.comm_stream.cuda_stream)
            else:
                self._lib.vitae_ddp_allreduce_bucket(self.flat.data_ptr() + 4 * s, e - s, 0, cs, self.comm_stream.cuda_stream)
            self.pending.append((None, s, e, bucket))

    def _join(self):
        self._lib.vitae_ddp_wait(torch.cuda.current_stream(self.device).cuda_stream, self.comm_stream.cuda_stream)

    def wait(self, copy_back: bool = True):
        if self.pending:
            self._join()
        for _, s, e, _b in self.pending:
            if self.wire is not None and copy_back:
                self.flat[s:e].copy_(self.wire[s:e])
        self.pending.clear()

    def wait_bucket(self, bucket: int, copy_back: bool = True):
        """Buckets complete in launch order on the one communication stream: joining it covers this bucket."""
        mine = [p for p in self.pending if p[3] == bucket]
        if mine:
            self._join()
        for _, s, e, _b in mine:
            if self.wire is not None and copy_back:
                self.flat[s:e].copy_(self.wire[s:e])
        self.pending = [p for p in self.pending if p[3] != bucket]


def allreduce_mean_now(reducer: "GradBucketReducer"):
    """Generic (non-fused) route: all-reduce every bucket and leave the MEAN of the per-rank gradients in the arena.  The
    fused step folds 1/world into the loss multipliers instead; here the gradients were produced unscaled, so they are
    divided after the SUM."""
    if reducer is None or not reducer.active:
        return
    for b in range(len(reducer.ranges)):
        reducer.launch(b)
    reducer.wait(copy_back=True)
    ws = reducer.world_size
    if ws > 1:
        reducer.flat.mul_(1.0 / ws)


def engine_bucket_ranges(engine) -> List[Tuple[int, int]]:
    """[decoder+predictor matrices (one or two buckets: ``engine.dec_cut``), encoder chunks from the top (the last one down to offset 0, i.e. with the patch
    embedding), tokens+vectors] as element ranges of ``engine.grads`` — the completion order of
    ``HipMAEEngine.train_phase``'s backward phases."""
    lay = engine.layout
    dec0 = lay['decoder_embed.weight'][0]
    cut = getattr(engine, 'dec_cut', 0)
    if cut > 0:     # two decoder buckets: [top blocks, decoder_pred, predictor], then [decoder_embed, bottom blocks]
        mid = lay[f'decoder_blocks.{cut}.attn.qkv.weight'][0]
        out = [(mid, engine.tok_off), (dec0, mid)]
    else:
        out = [(dec0, engine.tok_off)]
    top = dec0
    bounds = engine.enc_chunk_bounds()
    for i, (hi, lo) in enumerate(bounds):
        start = 0 if i == len(bounds) - 1 else lay[f'blocks.{lo}.attn.qkv.weight'][0]
        out.append((start, top))
        top = start
    out.append((engine.tok_off, engine.n_total))
    return out


def broadcast_parameters(engine, src: int = 0, group=None):
    """Start from identical replicas (what DistributedDataParallel does at construction)."""
    if not is_distributed():
        return
    dist.broadcast(engine.params, src=src, group=group)
    for k, t in engine.buffers.items():
        if k.startswith('predictor.1.'):
            dist.broadcast(t, src=src, group=group)
