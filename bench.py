"""Headline benchmark: pre-training volumes/sec of the ViT-B/16^3 (contrastive) MAE step on MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full optimisation step of BASELINE config 2 (ViT-B/16^3 autoenc as the reference
scripts train it, i.e. ``contr_mae_vit_base_patch16``; synthetic BraTS-shape 96^3 x 4ch volumes,
batch 4 per GPU, mask 0.75): forward (both views) + loss chain + backward + global grad norm +
AdamW, with the gradient all-reduce over RCCL at N > 1.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line (contract in the task statement) carrying ``roofline`` (the
GEMM kernel family, timed with HIP events on the launch stream in instrumented steps right after
the timed region) and, at N = 1, ``cpu_baseline`` (the oracle's CPU step on the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VOL, CH, PATCH = 96, 4, 16
PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
ALGO_GFLOP_PER_VOL = {'contr': 136.3, 'mae': 90.8}   # BASELINE.md §4 (fwd+bwd, reference formulation)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=4, help='volumes per GPU per step (BASELINE config 2: 4)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--model', default='contr', choices=['contr', 'mae'])
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--grad-comm', default=None, choices=['fp32', 'bf16'],
                    help='wire dtype of the gradient all-reduce at >1 GPU (default: the compute precision)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary plain-MAE data point')
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--profile-steps', type=int, default=3, help='instrumented steps for the roofline block')
    return ap.parse_args()


def synthetic_batches(batch, rank, n_batches=4):
    """view2 ~ N(0,1), view1 = z-score(view2 + 0.1 N(0,1)); CPU generator 1234+rank."""
    out = []
    for i in range(n_batches):
        g = torch.Generator(device='cpu').manual_seed(1234 + rank + 1000 * i)
        v2 = torch.randn(batch, CH, VOL, VOL, VOL, generator=g)
        v1 = v2 + 0.1 * torch.randn(batch, CH, VOL, VOL, VOL, generator=g)
        dims = (2, 3, 4)
        v1 = (v1 - v1.mean(dim=dims, keepdim=True)) / v1.std(dim=dims, keepdim=True)
        out.append((v1.contiguous(), v2.contiguous()))
    return out


def cpu_baseline(args, batches, sd_cpu, noises):
    """The oracle (CPU restatement of the reference, validated against it in tests) timed on this
    host: full training steps of the same workload.  Also returns the first-step losses."""
    from oracle import mae_ref as R
    from oracle import train_ref as T
    cfg = R.vit_base_cfg(volume_size=(VOL,) * 3, patch_size=PATCH, in_chans=CH, contrastive=args.model == 'contr')
    cores = torch.get_num_threads()
    tr = T.RefTrainer(cfg, sd_cpu, lr=1e-4, weight_decay=0.05)
    first = None
    times = []
    for i in range(args.cpu_steps + 1):
        v1, v2 = batches[i % len(batches)]
        n1, n2 = noises[i % len(noises)]
        t0 = time.perf_counter()
        terms, _, _ = tr.step(v1, v2, n1, n2, lr=1e-4, mask_ratio=0.75, edge_map_weight=0.01, contr_weight=0.001)
        dt = time.perf_counter() - t0
        if i == 0:
            first = terms
        else:
            times.append(dt)
    sec = sum(times) / len(times)
    return {'value': args.batch / sec, 'unit': 'volumes/s', 'cores': cores, 'kind': 'port',
            'sample': f'{args.cpu_steps} timed full steps (fwd+loss+bwd+AdamW) of the same workload, batch {args.batch}, '
                      f'fp32, torch {torch.__version__} CPU, after 1 warm-up step; {sec:.2f} s/step'}, first


def big_batch_point(args, dev, model, eng, contr, big=32):
    """Secondary data point: the headline model at batch 32 per GPU (what 288 GB of HBM is for).  The step stops being
    launch-bound there; same kernels, same graph machinery, device-generated synthetic volumes of the same distribution."""
    g = torch.Generator(device=dev).manual_seed(77)
    batches = []
    for _ in range(2):
        v2 = torch.randn(big, CH, VOL, VOL, VOL, generator=g, device=dev)
        v1 = v2 + 0.1 * torch.randn(big, CH, VOL, VOL, VOL, generator=g, device=dev)
        v1 = (v1 - v1.mean(dim=(2, 3, 4), keepdim=True)) / v1.std(dim=(2, 3, 4), keepdim=True)
        batches.append((v1.contiguous(), v2))
    runner = model._step_runner(big, 0.75, True, False, not args.no_graph)
    warm, steps = 2 * len(batches) + 2, 10
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        v1, v2 = batches[i % len(batches)]
        runner.load(v1, v2 if contr else None)
        eng.optimizer_hparams(lr=1e-4)
        runner.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gf = ALGO_GFLOP_PER_VOL[args.model] * big
    return {'batch': big, 'value': round(big / dt, 2), 'unit': 'volumes/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': steps,
            'step_tflops': round(gf / dt / 1e3, 1)}


def plain_mae_point(args, dev, batches):
    """Secondary data point: the same step for the plain (non-contrastive) `mae_vit_base_patch16` — one view, 55-token
    encoder.  BASELINE config 2 says "autoenc"; the reference's pre-training scripts train the contrastive model, which
    is the headline workload here (two views per volume: strictly more work per volume)."""
    from vit_ae_plus_plus_amd.model import vit_autoenc as VA
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    from oracle import mae_ref as R
    cfg = R.vit_base_cfg(volume_size=(VOL,) * 3, patch_size=PATCH, in_chans=CH, contrastive=False)
    model = VA.mae_vit_base_patch16(volume_size=VOL, in_chans=CH, patch_size=PATCH,
                                    args=argparse.Namespace(use_imagenet=False, perceptual_weight=0), precision=args.precision)
    model.load_state_dict(R.init_state_dict(cfg, seed=0))
    model = model.to(dev).train()
    eng = model._ensure_engine(dev)
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.05, betas=(0.9, 0.95))
    _ = opt.engine
    eng.set_loss_weights(0.01, 0.0, 1, 1)
    runner = model._step_runner(args.batch, 0.75, True, False, not args.no_graph)
    warm, steps = 2 * len(batches) + 2, 15      # each device batch gets its in-place graph on second sight: before the clock
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        runner.load(batches[i % len(batches)][0], None)
        eng.optimizer_hparams(lr=1e-4)
        runner.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {'model': 'mae_vit_base_patch16 (plain MAE, one view per volume)', 'value': round(args.batch / dt, 2),
            'unit': 'volumes/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': steps}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if rank != 0:   # only rank 0 talks on stdout (the contract is ONE JSON line)
        sys.stdout = open(os.devnull, 'w')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_ddp = os.environ.get('VITAE_FORCE_DDP') == '1' and world == 1   # single-GPU check of the N>1 machinery
    if world > 1 or force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)

    from vit_ae_plus_plus_amd.model import vit_autoenc as VA
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    from oracle import mae_ref as R   # weights by state-dict from the oracle's seed-0 init

    contr = args.model == 'contr'
    cfg = R.vit_base_cfg(volume_size=(VOL,) * 3, patch_size=PATCH, in_chans=CH, contrastive=contr)
    sd_cpu = R.init_state_dict(cfg, seed=0)
    margs = argparse.Namespace(use_imagenet=False, perceptual_weight=0)
    ctor = VA.contr_mae_vit_base_patch16 if contr else VA.mae_vit_base_patch16
    model = ctor(volume_size=VOL, in_chans=CH, patch_size=PATCH, args=margs, precision=args.precision)
    model.load_state_dict(sd_cpu)
    model = model.to(dev).train()
    eng = model._ensure_engine(dev)
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.05, betas=(0.9, 0.95))
    _ = opt.engine
    grad_comm = args.grad_comm or args.precision
    model.enable_data_parallel(dev, force=force_ddp, comm_dtype=torch.bfloat16 if grad_comm == 'bf16' else None)
    eng.set_loss_weights(0.01, 0.001 if contr else 0.0, 1, world)

    cpu_batches = synthetic_batches(args.batch, rank)
    batches = [(a.to(dev), b.to(dev)) for a, b in cpu_batches]
    L = cfg.num_patches
    noises = [R.masking_noise(args.batch, L, seed=4321 + rank + i) for i in range(len(batches))]
    runner = model._step_runner(args.batch, 0.75, True, False, not args.no_graph)

    def step(i):
        v1, v2 = batches[i % len(batches)]
        runner.load(v1, v2 if contr else None)     # device-resident batch: staged once, read in place when it comes back
        eng.optimizer_hparams(lr=1e-4)
        runner.run()

    # first step with the CPU-generated noise of step 0 for the parity report
    model.set_masking_noise(*(noises[0] if contr else noises[0][:1]))
    step(0)
    torch.cuda.synchronize()
    first_gpu = eng.losses.cpu().tolist()
    # graph priming (setup, like a compile step): a device batch is staged on first sight and gets a graph on its own
    # addresses on second sight, so every batch is shown twice before the W warm-up steps and the clock
    for i in range(1, 2 * len(batches)):
        step(i)
    for i in range(max(args.warmup, 1)):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    last = eng.losses.cpu().tolist()
    ms = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed

    # ---- roofline of the dominant kernel (the MFMA GEMM kernels of csrc/gemm_glds.hip / gemm_bf16.hip / gemm.hip):
    # instrumented eager steps right after the timed region, HIP events (torch.cuda.Event on the launch
    # stream = torch's current stream) around every GEMM launch.  A spin kernel is queued first so the host
    # enqueues the whole step while the GPU is still busy: event deltas then contain no host-launch gaps.
    roof = None
    if rank == 0 and args.profile_steps > 0:
        eager = model._step_runner(args.batch, 0.75, True, False, False)
        recs = []
        for i in range(args.profile_steps):
            v1, v2 = batches[i % len(batches)]
            eager.load(v1, v2 if contr else None)
            eng.optimizer_hparams(lr=1e-4)
            torch.cuda._sleep(200_000_000)
            eng.gemm_timer = []
            if world > 1:   # keep collectives matched: other ranks idle here, so time phases locally only
                for k in range(eng.N_PHASES - 1):
                    eager._phase(k)
            else:
                eager.run()
            torch.cuda.synchronize()
            recs += eng.gemm_timer
            eng.gemm_timer = None
        # calibration of the event pair itself: the same record / tiny kernel / record pattern, queued behind a spin
        # kernel like the steps.  A 16-byte fill runs for ~2 us (rocprofv3: the launch floor of this box); whatever the
        # events report beyond that is command-processor time around the kernel, not kernel time, and is removed.
        from vit_ae_plus_plus_amd._abi import lib as _lib
        scratch = torch.zeros(64, device=dev)
        torch.cuda._sleep(50_000_000)
        cal = []
        for _ in range(64):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.vitae_memset_zero(scratch.data_ptr(), 16, torch.cuda.current_stream(dev).cuda_stream)
            b.record()
            cal.append((a, b))
        torch.cuda.synchronize()
        null_ms = sorted(x.elapsed_time(y) for x, y in cal)[len(cal) // 2]
        overhead_ms = max(0.0, null_ms - 0.002)
        fam = {}
        for a, b, f, tag in recs:
            d = fam.setdefault(tag, [0.0, 0.0, 0, 0.0])
            raw = a.elapsed_time(b)
            d[0] += max(raw - overhead_ms, 0.25 * raw); d[1] += f; d[2] += 1; d[3] += raw
        tot_ms = sum(d[0] for d in fam.values())
        tot_fl = sum(d[1] for d in fam.values())
        dom = max(fam, key=lambda k: fam[k][0])          # the kernel the step spends most GEMM time in
        d_ms, d_fl, d_n, d_raw = fam[dom]
        ach = d_fl / (d_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        kname = {'glds_pair': 'gemm_glds_pair_kernel<64,64,64,64> (csrc/gemm_glds.hip: dgrad + wgrad of one Linear per launch)',
                 'glds': 'gemm_glds_kernel<64,64,..> (csrc/gemm_glds.hip)',
                 'glds_wide': 'gemm_glds_kernel<64,128,..> (csrc/gemm_glds.hip)',
                 'glds_pair_wide': 'gemm_glds_pair_kernel<..,64,128> (csrc/gemm_glds.hip)',
                 'glds_dgrad': 'gemm_glds_kernel<64,64,true,false> (csrc/gemm_glds.hip: dgrad of one Linear)',
                 'glds_wgrad_group': 'gemm_glds_group_kernel (csrc/gemm_glds.hip: the four weight gradients of a block in one launch)',
                 'other': ('gemm_bf16_kernel / gemm_bf16_pair_kernel (csrc/gemm_bf16.hip)' if args.precision == 'bf16'
                           else 'gemm_kernel<0,..> (csrc/gemm.hip)')}.get(dom, dom)
        traffic, tnote = None, None   # PMC counters cannot be read in-process: last committed rocprofv3 --pmc result
        tfile = os.path.join(ROOT, 'profiles', 'round1_gemm_traffic.json')
        if args.precision == 'bf16' and args.batch == 4 and os.path.exists(tfile):
            tj = json.load(open(tfile))
            traffic = tj.get('bytes_per_launch', {}).get(dom)
            tnote = 'HBM-side bytes per launch of that kernel (2*FETCH_SIZE + WRITE_SIZE, profiles/round1_gemm_traffic.json)'
        roof = {'bound': 'mfma', 'kernel': kname, 'achieved': round(ach, 2),
                'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': traffic,
                'traffic_unit': tnote,
                'launches_per_step': d_n / args.profile_steps, 'gflop_per_launch': round(d_fl / d_n / 1e9, 3),
                'avg_launch_us': round(d_ms * 1e3 / d_n, 2),
                'avg_launch_us_events_raw': round(d_raw * 1e3 / d_n, 2), 'event_pair_overhead_us': round(overhead_ms * 1e3, 2),
                'gemm_family': {'launches_per_step': len(recs) / args.profile_steps,
                                'gflop_per_step': round(tot_fl / args.profile_steps / 1e9, 2),
                                'ms_per_step': round(tot_ms / args.profile_steps, 3),
                                'tflops': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                'by_kernel_ms_per_step': {k: round(v[0] / args.profile_steps, 3) for k, v in fam.items()}},
                'step_frac_of_peak': round(world * args.batch * ALGO_GFLOP_PER_VOL[args.model] * 1e9 / (ms * 1e-3) / 1e12 / (peak * world), 4)}
    if world > 1:
        dist.barrier()

    also, also_big = None, None
    if rank == 0 and world == 1 and not force_ddp and not args.no_extra and args.batch < 32:
        try:
            also_big = big_batch_point(args, dev, model, eng, contr)
        except Exception as e:   # a secondary point must never cost the headline line
            also_big = {'error': repr(e)[:200]}
    if rank == 0 and world == 1 and contr and not force_ddp and not args.no_extra:
        try:
            also = plain_mae_point(args, dev, batches)
        except Exception as e:
            also = {'error': repr(e)[:200]}

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, first_cpu = cpu_baseline(args, cpu_batches, sd_cpu, noises)
        ref_total, ref_recon = first_cpu['loss'], first_cpu['reconstruction_loss']
        got_total, got_recon = first_gpu[0] + (first_gpu[4] if contr else 0.0), first_gpu[2]
        parity = {'recon_loss_gpu': got_recon, 'recon_loss_cpu_oracle': ref_recon,
                  'recon_rel_err': abs(got_recon - ref_recon) / abs(ref_recon),
                  'total_rel_err': abs(got_total - ref_total) / abs(ref_total), 'precision': args.precision}

    if rank == 0:
        out = {'metric': 'pretrain volumes/sec (96^3x4ch, mask 0.75) at 1/2/4/8 MI355X + recon-loss parity',
               'value': round(value, 2), 'unit': 'volumes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': args.precision, 'data': 'synthetic',
               'config': {'workload': f'ViT-B/16^3 {"contrastive " if contr else ""}MAE full optimisation step '
                                      f'(fwd+loss+bwd+grad-norm+AdamW), synthetic BraTS-shape 96^3x4ch, batch '
                                      f'{args.batch}/GPU, mask 0.75 (BASELINE config 2{" / 3" if world > 1 else ""})',
                          'global_batch': world * args.batch, 'parallelism': f'dp{world}',
                          'hip_graph': not args.no_graph,
                          'grad_allreduce': (f'{grad_comm}, {len(model._reducer.ranges)} buckets' if model._reducer is not None else None), 'final_losses': [round(x, 6) for x in last[:6]]},
               'roofline': roof, 'cpu_baseline': cpu}
        if also:
            out['config']['also'] = also
        if also_big:
            out['config']['also_batch32'] = also_big
        if parity:
            out['parity'] = parity
    if world > 1 or force_ddp:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio: drain that first so the JSON line is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
