"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mvlpt_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    path = os.path.join(ROOT, "mvlpt_amd", "libmvlpt_hip.so")
    if not os.path.isfile(path):
        import __graft_entry__ as g
        g.build()
    return path


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mvlpt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvlpt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvlpt_hip.h but not exported"


def test_python_binding_covers_header(lib_path):
    from mvlpt_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib.mvlpt_version().decode().startswith("mvlpt_hip")


def test_create_fails_loudly_without_gpu(lib_path):
    """No CPU fallback: on a box without a HIP device the handle cannot even be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mvlpt_amd import _lib
    a = _lib.MvlptArch(32, 16, 128, 2, 2, 77, 128, 2, 2, 128, _lib.DT_F16)
    h = ctypes.c_void_p()
    rc = _lib.lib.mvlpt_create(ctypes.byref(a), ctypes.byref(h))
    assert rc < 0 and "no HIP device" in _lib.last_error(None)
    from mvlpt_amd.engine import Engine
    from mvlpt_amd.weights import ARCHS
    with pytest.raises(RuntimeError):
        Engine(ARCHS["tiny"])


def test_arch_validation(lib_path):
    from mvlpt_amd import _lib
    h = ctypes.c_void_p()
    bad = _lib.MvlptArch(32, 16, 100, 2, 2, 77, 128, 2, 2, 128, _lib.DT_F16)   # width != heads*64
    assert _lib.lib.mvlpt_create(ctypes.byref(bad), ctypes.byref(h)) == -4
    bad = _lib.MvlptArch(32, 16, 128, 2, 2, 77, 128, 2, 2, 128, 0)             # fp32 compute is not offered
    assert _lib.lib.mvlpt_create(ctypes.byref(bad), ctypes.byref(h)) == -1
