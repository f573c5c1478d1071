"""The C-ABI library must load (no GPU needed for that) and export every symbol that
include/roms_hip.h declares; without a GPU, create must fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "roms_hip.h")).read()
    return sorted(set(re.findall(r"\b(roms_hip_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from roms_amd import build, hiplib
    build.build_hip()
    L = ctypes.CDLL(hiplib.DEFAULT_LIB)
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), s
    assert L.roms_hip_abi_version() == 5
    assert sorted(hiplib.EXPORTS) == syms


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from roms_amd import hiplib
    from tests import util
    cs = util.case_for("upwelling_small")
    g = util.load_init("upwelling_small", 3)
    with pytest.raises(hiplib.RomsHipError) as e:
        util.make_hip(cs, g)
    assert "no HIP device" in str(e.value) or "exit_flag=2" in str(e.value)
