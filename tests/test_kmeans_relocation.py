"""Empty-cluster relocation (scikit-learn _relocate_empty_clusters_dense): degenerate count matrices (duplicate rows,
k close to the number of distinct points) through the kernels' logic (emulation build) against the oracle.  The
oracle's rule restates NumPy's generic arg-introselect, i.e. the reference's locked NumPy 1.24 (no SIMD sort)."""
import numpy as np

import oracle.from_msa_oracle as orc
from tests.emu.backend import EmuBackend
from tests.kmeans_direct import run_kmeans_fits


def degenerate_fits(seed, n):
    rng = np.random.default_rng(seed)
    fits = []
    while len(fits) < n:
        D = int(rng.integers(4, 14)); V = int(rng.integers(1, 5)); k = int(rng.integers(2, min(D, 10)))
        M = rng.integers(0, 3, (D, V)).astype(np.float64)
        lab, dbg = orc.kmeans_fit_predict(M, k, want_debug=True)
        if dbg["flags"] & 1:
            fits.append(dict(shape=[D, V], counts_i16_hex=M.astype("<i2").tobytes().hex(), k=k, labels=lab.tolist(),
                             inertia=float(dbg["inertia"]).hex(), n_iter=dbg["n_iter"]))
    return fits


def check(backend, fits, path="global", n_slots=3):
    got = run_kmeans_fits(backend, fits, path=path, n_slots=n_slots)
    for g, f in zip(got, fits):
        assert g["status"] & 1 and not g["status"] & 2
        assert g["labels"] == f["labels"]
        assert g["inertia_hex"] == f["inertia"]
        assert g["n_iter"] == f["n_iter"]


def test_relocation_matches_oracle():
    check(EmuBackend(), degenerate_fits(7, 60))


def test_relocation_matches_oracle_persistent_workgroups():
    fits = degenerate_fits(7, 60)
    check(EmuBackend(), fits, path="wave")                        # a wavefront per fit, restart state in LDS
    check(EmuBackend(), fits, path="lds")                         # a workgroup per fit, the state of all restarts in LDS (round 6)
    check(EmuBackend(), fits, path="fit", n_slots=2)              # many fits per workgroup, one scratch slot each
    check(EmuBackend(), fits, path="split")                       # a workgroup per restart, then the selection
    check(EmuBackend(), fits[:12], path="wide")                   # the same with 1 024 threads per restart (big fits)
    # ... and with the counts read as bytes from the caller's byte matrix (what a big fit's wide workgroups do: test-only build whose pool holds none)
    check(EmuBackend(defines=("MPRG_TEST_WIDE_GLOBAL",), tag="_wideglobal"), fits[:12], path="wide-bytes")
    # ... and without the sample-sample tables of the seeding (mprg_kmeans_prepare_big, with_tables = 0): their elements on demand, from bytes and doubles
    check(EmuBackend(defines=("MPRG_TEST_WIDE_GLOBAL",), tag="_wideglobal"), fits[12:24], path="wide-stats")
    check(EmuBackend(), fits[24:32], path="wide-stats")
    check(EmuBackend(), fits, path="fit", n_slots=512)


def test_fits_with_many_samples():
    """Count matrices with more samples than the pan-genome shapes of the golden fits (several 8-wide operand blocks)."""
    rng = np.random.default_rng(3)
    fits = []
    for D, V, k in ((70, 9, 3), (90, 5, 2), (130, 7, 10), (66, 30, 6)):
        centres = rng.integers(0, 6, (4, V))
        M = (centres[rng.integers(0, 4, D)] + rng.integers(0, 2, (D, V))).astype(np.float64)
        lab, dbg = orc.kmeans_fit_predict(M, k, want_debug=True)
        fits.append(dict(shape=[D, V], counts_i16_hex=M.astype("<i2").tobytes().hex(), k=k, labels=lab.tolist(),
                         inertia=float(dbg["inertia"]).hex(), n_iter=dbg["n_iter"]))
    for path, slots in (("global", 0), ("one-launch", 0), ("wave", 0), ("small", 0), ("lds", 0), ("fit", 1), ("fit", 64)):
        got = run_kmeans_fits(EmuBackend(), fits, path=path, n_slots=slots)
        for g, f in zip(got, fits):
            assert not g["status"] & 2
            assert g["labels"] == f["labels"] and g["inertia_hex"] == f["inertia"] and g["n_iter"] == f["n_iter"]


def test_relocation_with_wide_matrices():
    """More than 128 features: the tolerance (np.var(...).mean()) and the relocation distances take NumPy's pairwise-sum
    recursion, which the kernels run on a stack in LDS."""
    rng = np.random.default_rng(11)
    fits = []
    for V in (150, 300, 700, 1100):
        for _ in range(200):
            D = int(rng.integers(6, 13)); n_distinct = int(rng.integers(2, 4)); k = int(rng.integers(n_distinct + 1, min(D, 6) + 1))
            base = rng.integers(0, 3, (n_distinct, V))
            M = base[rng.integers(0, n_distinct, D)].astype(np.float64)
            lab, dbg = orc.kmeans_fit_predict(M, k, want_debug=True)
            if dbg["flags"] & 1:
                fits.append(dict(shape=[D, V], counts_i16_hex=M.astype("<i2").tobytes().hex(), k=k, labels=lab.tolist(),
                                 inertia=float(dbg["inertia"]).hex(), n_iter=dbg["n_iter"]))
                break
    assert len(fits) >= 3
    check(EmuBackend(), fits)
    check(EmuBackend(), fits, path="one-launch")
    check(EmuBackend(), fits, path="wave")
    check(EmuBackend(), fits, path="small")
    check(EmuBackend(), fits, path="lds")
    check(EmuBackend(), fits, path="fit", n_slots=2)


def test_fits_outside_the_lds_count_form():
    """The restart kernel keeps a fit's counts as bytes in LDS when they fit (km_euclid_xl); counts above 255 and matrices
    beyond the pool take the global-memory form — same answers."""
    rng = np.random.default_rng(5)
    fits = []
    for D, V, k, top in ((24, 40, 3, 400), (120, 130, 4, 6), (14, 23, 5, 256), (31, 61, 2, 3)):
        centres = rng.integers(0, top, (4, V))
        M = (centres[rng.integers(0, 4, D)] + rng.integers(0, 2, (D, V))).astype(np.float64)
        lab, dbg = orc.kmeans_fit_predict(M, k, want_debug=True)
        fits.append(dict(shape=[D, V], counts_i16_hex=M.astype("<i2").tobytes().hex(), k=k, labels=lab.tolist(),
                         inertia=float(dbg["inertia"]).hex(), n_iter=dbg["n_iter"]))
    for path, slots in (("global", 0), ("global-nocounts", 0), ("one-launch", 0), ("wave", 0), ("small", 0), ("lds", 0), ("fit", 2)):
        got = run_kmeans_fits(EmuBackend(), fits, path=path, n_slots=slots)
        for g, f in zip(got, fits):
            assert not g["status"] & 2
            assert g["labels"] == f["labels"] and g["inertia_hex"] == f["inertia"] and g["n_iter"] == f["n_iter"]
