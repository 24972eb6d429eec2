"""np.argpartition as scikit-learn's empty-cluster relocation uses it: the generic arg-introselect INCLUDING its median-of-medians
fallback.  tests/golden/argpartition.json (oracle/tools/gen_argpartition_killer.py): inputs built with McIlroy's adversary so
that the fallback runs 4-9 times, answers of the REAL np.argpartition (x86-simd-sort dispatch disabled = the generic code of the
reference's locked NumPy 1.24).  Oracle (C) and kernel source (emulation here, HIP in test_gpu_parity.py) against them."""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    with open(os.path.join(HERE, "golden", "argpartition.json")) as fh:
        return json.load(fh)["cases"]


def check_device(be):
    n_fallback = 0
    for c in cases():
        v = np.asarray(c["values"], np.float64)
        d_v, d_perm, d_ok = be.upload(v), be.empty(4 * c["n"]), be.zeros(16)
        be.call("mprg_argpartition", be.ptr(d_v), be.ptr(d_perm), c["n"], c["kth"], be.ptr(d_ok), be.stream)
        assert int(be.download(d_ok, np.int32, 1)[0]) == 1
        assert be.download(d_perm, np.int32, c["n"]).tolist() == c["argpartition"], (c["n"], c["kth"])
        n_fallback += c["fallbacks"] or 0
    assert n_fallback >= 30          # the fixture does reach the fallback
    return len(cases())


def test_oracle_selection_matches_numpy():
    import oracle.from_msa_oracle as orc
    lib = ctypes.CDLL(orc.build_kmeans_lib())
    lib.mprg_oracle_argpartition.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long]
    for c in cases():
        v = np.asarray(c["values"], np.float64)
        perm = np.arange(c["n"], dtype=np.int64)
        lib.mprg_oracle_argpartition(v.ctypes.data, perm.ctypes.data, c["n"], c["kth"])
        assert perm.tolist() == c["argpartition"]


def test_kernel_selection_matches_numpy():
    from tests.emu.backend import EmuBackend
    assert check_device(EmuBackend()) == 9
    rng = np.random.default_rng(5)          # and random inputs against the oracle (ties, small kth, kth = n - 1)
    import oracle.from_msa_oracle as orc
    lib = ctypes.CDLL(orc.build_kmeans_lib())
    lib.mprg_oracle_argpartition.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long]
    be = EmuBackend()
    for t in range(300):
        n = int(rng.integers(1, 200))
        v = rng.integers(0, rng.integers(2, 30), n).astype(np.float64) if t % 2 else rng.random(n)
        kth = int(rng.integers(0, n)) if t % 5 else n - 1
        want = np.arange(n, dtype=np.int64)
        lib.mprg_oracle_argpartition(v.ctypes.data, want.ctypes.data, n, kth)
        d_v, d_perm, d_ok = be.upload(v), be.empty(4 * n), be.zeros(16)
        be.call("mprg_argpartition", be.ptr(d_v), be.ptr(d_perm), n, kth, be.ptr(d_ok), be.stream)
        assert be.download(d_perm, np.int32, n).tolist() == want.tolist(), (n, kth)
