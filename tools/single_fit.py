"""GPU-box helper: latency of ONE KMeans fit (one workgroup) per k, for a problem shaped like the largest of a config-C
batch (D ~ 48 distinct sequences, V ~ 480 distinct 7-mers).  Shows what bounds the tail of a k_kmeans_restart launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from make_prg_amd.backend import HipBackend
from kmeans_direct import run_kmeans_fits

rng = np.random.default_rng(5)
D, L, K = int(sys.argv[1]) if len(sys.argv) > 1 else 48, int(sys.argv[2]) if len(sys.argv) > 2 else 420, 7
founders = rng.integers(0, 4, (4, L))
seqs = []
for i in range(D):
    s = founders[i % 4].copy()
    m = rng.random(L) < 0.01
    s[m] = rng.integers(0, 4, int(m.sum()))
    seqs.append(s)
kmers = {}
rows = []
for s in seqs:
    c = {}
    for p in range(L - K + 1):
        key = tuple(s[p:p + K])
        idx = kmers.setdefault(key, len(kmers))
        c[idx] = c.get(idx, 0) + 1
    rows.append(c)
V = len(kmers)
X = np.zeros((D, V), np.int16)
for i, c in enumerate(rows):
    for j, n in c.items():
        X[i, j] = n
print("D", D, "V", V)
import ctypes
timing = os.environ.get("MPRG_TIMING_LIB")
be = HipBackend(0, lib_path=timing) if timing else HipBackend(0)
if timing:
    be.lib.mprg_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
be.profile = {}
for rep in range(2):
    for k in range(2, 11):
        be.profile.clear()
        res = run_kmeans_fits(be, [dict(k=k, shape=(D, V), counts_i16_hex=X.astype("<i2").tobytes().hex())])
        e0, e1, _ = be.profile["mprg_kmeans_restarts"][-1]
        p0, p1, _ = be.profile["mprg_kmeans_prepare"][-1]
        if rep and timing:
            out = (ctypes.c_ulonglong * 16)()
            be.lib.mprg_debug_phase_cycles(out, 1)
            c = np.array(list(out), dtype=np.float64)
            print("   phase % :", " ".join(f"{100 * v / c.sum():.0f}" for v in c[:10]), " Mcycles", round(c.sum() / 1e6, 2))
        elif timing:
            be.lib.mprg_debug_phase_cycles(None, 1)
        if rep:
            print("k", k, "restart ms", round(e0.elapsed_time(e1), 3), "prepare ms", round(p0.elapsed_time(p1), 3), "n_iter(best)", res[0]["n_iter"])
