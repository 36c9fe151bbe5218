"""Builds ``libvitae_hip.so`` (the C-ABI HIP library, gfx950 only) in-tree with hipcc.

    python -m vit_ae_plus_plus_amd.build [--force]

No torch headers are involved: the library is plain HIP behind ``include/vitae_hip.h``.  Objects are
cached per source under ``csrc/_obj`` keyed on mtime so incremental rebuilds take seconds.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libvitae_hip.so')
SOURCES = ['gemm.hip', 'gemm_bf16.hip', 'gemm_glds.hip', 'gemm_bt.hip', 'norm.hip', 'attention.hip', 'attention_mfma.hip', 'tokens.hip', 'loss.hip', 'loss_fused.hip',
           'optim.hip', 'input.hip', 'ddp.hip', 'percep.hip']
# -amdgpu-kernarg-preload-count=16: the first 16 dwords of a kernel's scalar / pointer arguments arrive in SGPRs with the dispatch instead
# of through a cold scalar load at the top of every launch (round 6: the batch-4 step 3.662 -> 3.636 ms, alternating A/B; kernels that take
# their arguments as ONE by-value struct — the LDS-DMA GEMM family's GArgs — are not covered by the option)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-mllvm', '-amdgpu-kernarg-preload-count=16',
         '-I', os.path.join(ROOT, 'include'), '-I', CSRC] + os.environ.get('VITAE_HIPCC_FLAGS', '').split()


# Per-source flags.  attention_mfma.hip: MFMA accumulators in VGPRs — its softmax touches every accumulator register of every tile,
# and with the accumulators in AGPRs (hipcc's default at this register count) each touch was a v_accvgpr_read / _write: 80 of the
# 170 VALU instructions per 32 x 32 tile of the forward (ISA), 6-12 % of the kernels' time (tools/attn_bench.py).  The GEMM files
# measured neutral to slightly slower with it (their accumulators are only read once, in the epilogue) and keep the default.
EXTRA_FLAGS = {'attention_mfma.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _newer(a: str, deps) -> bool:
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'glds_tiles.hpp'), os.path.join(CSRC, 'glds_gemm.hpp'), os.path.join(ROOT, 'include', 'vitae_hip.h')]
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or not _newer(obj, [src] + headers + [os.path.abspath(__file__)]):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), '-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed:\n{r.stdout}\n{r.stderr}')
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not _newer(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs, '-ldl'])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
