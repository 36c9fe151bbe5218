"""ctypes binding of ``libvitae_hip.so``; ``include/vitae_hip.h`` is the single source of truth.

The prototypes are parsed from the header, so the Python side can never drift from the C ABI.
There is NO fallback: if the library is missing or a launcher returns non-zero, this raises.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

# torch must come first: it ships its own libamdhip64.so and libvitae_hip.so has to bind to THAT HIP
# runtime instance (same SONAME -> the loader reuses the one already mapped).  Loading our library first
# would map /opt/rocm's copy and leave the process with a runtime that does not know torch's streams and
# allocations (seen as VITAE_ERR_LAUNCH from the first hipMemsetAsync).
import torch  # noqa: F401  (import order matters)

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
HEADER = os.path.join(ROOT, 'include', 'vitae_hip.h')
LIB_PATH = os.environ.get('VITAE_HIP_LIB') or os.path.join(PKG, 'libvitae_hip.so')   # the override: kernel experiments (tools/)

_SYNC_LAUNCHES = os.environ.get('VITAE_SYNC_LAUNCHES') == '1'
_ERRORS = {-1: 'VITAE_ERR_INVALID_ARG', -2: 'VITAE_ERR_UNSUPPORTED_SHAPE', -3: 'VITAE_ERR_LAUNCH'}


class VitaeError(RuntimeError):
    pass


def parse_header(path: str = HEADER) -> Tuple[Dict[str, Tuple[str, List[str]]], Dict[str, int]]:
    """-> ({function: (return type, [arg types])}, {macro: int value})"""
    text = open(path).read()
    text_nc = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define\s+(VITAE_\w+)\s+(-?\d+)\b', text_nc)}
    protos: Dict[str, Tuple[str, List[str]]] = {}
    for m in re.finditer(r'\b(int|long|const char\*)\s+(vitae_\w+)\s*\(([^)]*)\)\s*;', text_nc):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        kinds: List[str] = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    kinds.append('ptr')
                elif a.startswith('long long'):
                    kinds.append('longlong')
                elif a.startswith('long'):
                    kinds.append('long')
                elif a.startswith('int'):
                    kinds.append('int')
                elif a.startswith('float'):
                    kinds.append('float')
                elif a.startswith('double'):
                    kinds.append('double')
                else:
                    raise ValueError(f'unparsed argument {a!r} of {name}')
        protos[name] = (ret, kinds)
    return protos, consts


_CT = {'ptr': ctypes.c_void_p, 'longlong': ctypes.c_longlong, 'long': ctypes.c_long, 'int': ctypes.c_int,
       'float': ctypes.c_float, 'double': ctypes.c_double}
_RET = {'int': ctypes.c_int, 'long': ctypes.c_long, 'const char*': ctypes.c_char_p}

PROTOS, CONSTS = parse_header()


class _Lib:
    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise VitaeError(
                f'{LIB_PATH} is missing: the HIP library has not been built. Run '
                f'`python -m vit_ae_plus_plus_amd.build` (needs hipcc); there is no CPU fallback.')
        dll = ctypes.CDLL(LIB_PATH)
        for name, (ret, kinds) in PROTOS.items():
            fn = getattr(dll, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = _RET[ret]
            fn.argtypes = [_CT[k] for k in kinds]
        if dll.vitae_abi_version() != CONSTS['VITAE_ABI_VERSION']:
            raise VitaeError('libvitae_hip.so ABI version does not match include/vitae_hip.h; rebuild')
        self._dll = dll
        return dll

    def __getattr__(self, name):
        dll = self.load()
        fn = getattr(dll, name)
        if PROTOS[name][0] != 'int' or name in _VALUE_RETURNING or ('pick_split_k' in name):
            return fn

        def checked(*args):
            rc = fn(*args)
            if rc != 0:
                raise VitaeError(f'{name} failed: {_ERRORS.get(rc, rc)}')
            if _SYNC_LAUNCHES:      # measurement only (tools/in_step_tax.py): every launch alone on an idle chip
                torch.cuda.synchronize()
            return rc

        checked.__name__ = name
        setattr(self, name, checked)
        return checked


# int-returning entry points whose result is a value, not a status
_VALUE_RETURNING = {'vitae_abi_version', 'vitae_sdpa_bwd_fused_fits', 'vitae_loss_fwd_bwd_supported', 'vitae_target_edge_supported', 'vitae_ddp_available',
                    'vitae_ddp_world_size', 'vitae_gemm_glds_bt_choice', 'vitae_layernorm_bwd_part_records'}

lib = _Lib()


def const(name: str) -> int:
    return CONSTS[name]
