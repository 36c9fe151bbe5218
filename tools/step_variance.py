"""Where does the process-to-process spread of the batch-4 step (4.37 .. 4.54 ms) come from?  Within one process the step is stable to
0.3 % across re-captures of the graph and across fresh model instances; this prints what differs between processes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda', 0)
pad = int(os.environ.get('VITAE_PROBE_PAD_MB', '0'))
keep = torch.empty(pad << 20, dtype=torch.uint8, device=dev) if pad else None
batches = bench.device_batches(4, dev)
model, _, eng = bench.build_model('contr', 'bf16', dev)
eng.set_loss_weights(0.01, 0.001, 1, 1)
t = bench.run_steps(model, eng, batches, True, 4, True, 10, 80) * 1e3
ptrs = {k: hex(v.data_ptr()) for k, v in (('params', eng.params), ('grads', eng.grads), ('m', eng.opt_state['exp_avg']), ('v', eng.opt_state['exp_avg_sq']), ('p16', eng.params16), ('ws16', eng.ws16))}
print(f'step {t:.3f} ms  pad {pad} MB  {ptrs}', flush=True)
