"""The C-ABI library builds for gfx950 on a CPU-only machine, loads, and exports every symbol that
include/mtip_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
LIB = os.path.join(ROOT, 'xframe_amd', 'csrc', 'libmtip_hip.so')
HEADER = os.path.join(ROOT, 'include', 'mtip_hip.h')


@pytest.fixture(scope='module')
def lib():
    r = subprocess.run(['make', '-C', os.path.dirname(LIB), '-j5'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return ctypes.CDLL(LIB)


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mtip(?:2d)?_[a-z0-9_]+)\s*\(', txt)))


def test_every_declared_symbol_is_exported(lib):
    names = header_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_covers_the_header():
    from xframe_amd.fxs import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == header_symbols()


def test_no_device_is_reported_not_faked(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    lib.mtip_device_count.restype = ctypes.c_int
    assert lib.mtip_device_count() == 0
    from xframe_amd.fxs import _lib
    from xframe_amd.fxs.engine import Engine
    with pytest.raises(_lib.MtipError):
        Engine({'grid': {'n_radial_points': 8, 'max_order': 2}}, None, max_q=1.0)


def test_gfx950_code_object_present():
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', LIB], capture_output=True, text=True)
    txt = subprocess.run(['strings', LIB], capture_output=True, text=True).stdout
    assert 'gfx950' in txt
