"""Pinned four-step trajectory (tests/golden/vitb_b4.npz: the reference model's own losses) of the benchmarked bf16 route under a
two-plane weight selection, next to the batch-4 step time of the same selection:
    VITAE_W2=dec.fc1 python tools/w2_parity.py        VITAE_W2=decoder python tools/w2_parity.py        VITAE_W2= python tools/w2_parity.py
(engine._init_w2 reads VITAE_W2 when the engine is built.)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse(['--batch', '4', '--no-extra', '--no-cpu-baseline', '--steps', '40', '--warmup', '10'])
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
pt = bench.pinned_trajectory_parity(args, dev)
model, _, eng = bench.build_model('contr', 'bf16', dev)
eng.set_loss_weights(0.01, 0.001, 1, 1)
batches = bench.device_batches(4, dev)
ts = sorted(bench.run_steps(model, eng, batches, True, 4, True, 2 * len(batches) + 2, 60) for _ in range(int(os.environ.get('W2_REPS', '5'))))
dt = ts[len(ts) // 2]
print(f"VITAE_W2={os.environ.get('VITAE_W2', '<default>')!r}: {len(eng._w2)} two-plane tensors | worst rel err total {pt['worst_total_loss_rel_err']:.2e} "
      f"raw edge {pt['worst_raw_edge_rel_err']:.2e} recon {pt['worst_recon_loss_rel_err']:.2e} contr {pt['worst_contr_rel_err']:.2e} | "
      f"batch-4 step median {dt * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ({4 / dt:.1f} volumes/s)")
