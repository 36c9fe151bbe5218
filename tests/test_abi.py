"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/vitae_hip.h
declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from vit_ae_plus_plus_amd import _abi, build


@pytest.fixture(scope='module')
def libpath():
    return build.build(verbose=False)


def test_header_parses_every_prototype():
    text = open(_abi.HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    declared = set(re.findall(r'\b(vitae_\w+)\s*\(', text))
    assert declared == set(_abi.PROTOS), declared ^ set(_abi.PROTOS)
    assert len(declared) >= 30
    for name in ('vitae_linear_fwd', 'vitae_layernorm_bwd', 'vitae_sdpa_bwd', 'vitae_recon_loss_fwd',
                 'vitae_sobel_edge_bwd', 'vitae_gauss_blur_fwd', 'vitae_adamw_step', 'vitae_grad_sqnorm',
                 'vitae_random_masking', 'vitae_bn1d_relu_fwd', 'vitae_cosine_loss_bwd'):
        assert name in declared


def test_library_exports_all_declared_symbols(libpath):
    dll = ctypes.CDLL(libpath)
    for name in _abi.PROTOS:
        assert hasattr(dll, name), f'{name} declared in vitae_hip.h but not exported'
    lib = _abi.lib.load()
    assert lib.vitae_abi_version() == _abi.CONSTS['VITAE_ABI_VERSION']
    assert lib.vitae_build_arch() == b'gfx950'


def test_no_torch_types_in_abi():
    text = open(_abi.HEADER).read()
    code = re.sub(r'/\*.*?\*/', '', text, flags=re.S)            # declarations only: the comments cite reference files
    assert 'at::' not in code and 'torch' not in code.lower() and 'tensor' not in code.lower() and '#include <ATen' not in text
    for name, (ret, kinds) in _abi.PROTOS.items():
        assert set(kinds) <= {'ptr', 'int', 'long', 'longlong', 'float', 'double'}, name


def test_code_object_is_gfx950_only(libpath):
    # llvm-objdump --offloading EXTRACTS every bundle next to its input: work on a copy so nothing lands in the package
    # (in a directory whose name cannot be mistaken for an architecture in the tool's output)
    import shutil, tempfile
    tmp = tempfile.mkdtemp(prefix='vitae_codeobj_')
    copy = shutil.copy(libpath, os.path.join(tmp, 'lib.so'))
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', copy], capture_output=True, text=True)
    shutil.rmtree(tmp, ignore_errors=True)
    if out.returncode != 0:
        pytest.skip('llvm-objdump --offloading unavailable')
    archs = set(re.findall(r'gfx\w+', out.stdout))
    assert archs == {'gfx950'}, archs


def test_pure_host_entry_points(libpath):
    lib = _abi.lib
    # split-K heuristic is host arithmetic: few output tiles + long K -> split; many tiles -> 1
    assert lib.vitae_gemm_pick_split_k(432, 768, 16384) > 1
    assert lib.vitae_gemm_pick_split_k(868, 16384, 512) == 1
    assert lib.vitae_gemm_workspace_floats(100, 200, 4096, 4) == 4 * 100 * 200
    assert lib.vitae_gemm_workspace_floats(100, 200, 4096, 1) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_abi, 'LIB_PATH', str(tmp_path / 'nope.so'))
    fresh = _abi._Lib()
    with pytest.raises(_abi.VitaeError, match='no CPU fallback'):
        fresh.load()


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.abspath(_abi.__file__))
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dp, f)
