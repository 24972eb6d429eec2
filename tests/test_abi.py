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


def test_host_library_builds_without_the_hip_runtime():
    """libmprg_host.so: the host-only entry points (encoders, FASTA parser), same sources, no HIP dependency — what
    make_prg_amd.utils.native binds, so that parents of GPU worker processes never load a HIP runtime."""
    import subprocess
    from make_prg_amd.utils import native
    path = ge.build_host()
    lib = native.bind_host(ctypes.CDLL(path))
    for name in native.HOST_SIGNATURES:
        assert name in declared_symbols() and hasattr(lib, name)
    needed = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "hsa" not in needed
    out = np.empty(8, np.uint32)
    assert lib.mprg_prg_encode_host(b"AC 5 G 6 T 5 ", 13, out.ctypes.data) == 7 and out[:7].tolist() == [1, 2, 5, 3, 6, 4, 6]


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
    # the reference-shaped helper functions too: inputs of the kernels' domain need the device, nothing answers in its place
    from make_prg_amd import device
    from make_prg_amd.from_msa import cluster_sequences as cs
    device.set_backend(None)
    with pytest.raises(backend.MprgError):
        cs.sequences_are_one_reference_like(["ACGT", "ACGA"])
    with pytest.raises(backend.MprgError):
        cs.cluster_further([["ACGT", "ACGA"], ["TTTT"]])
