"""CPU-side checks of the boundary: the HIP shared object builds for gfx950, loads, and exports every symbol that
include/mprg.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np

import __graft_entry__ as ge
from make_prg_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mprg.h")).read()
    return sorted(set(re.findall(r"\b(mprg_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_builds_and_exports_the_declared_abi():
    lib = ctypes.CDLL(ge.build_hip())
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mprg.h but not exported"
    assert set(backend.SIGNATURES) == set(names)
    assert b"hip gfx950" in backend.bind(lib).mprg_version()


def test_host_rng_matches_numpy_randomstate():
    lib = backend.bind(ctypes.CDLL(ge.build_hip()))
    out = np.zeros(500)
    lib.mprg_random_sample_host(2, 500, out.ctypes.data)
    assert np.array_equal(out, np.random.RandomState(2).random_sample(500))


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(backend.MprgError):
        backend.HipBackend(0)
