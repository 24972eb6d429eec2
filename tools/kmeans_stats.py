"""GPU-box helper: statistics of the KMeans problems of a config-C batch (sizes, final k, iterations)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0)
eng = F.ForestEngine(be, 5, 7)
eng.load(msas)
log = []
orig = be.call


def call(name, *a, **k):
    if name == "mprg_kmeans_restarts":
        log.append(("restarts", a[1], a[2]))
    return orig(name, *a, **k)


be.call = call
# capture D, V per problem by wrapping download of V
orig_dl = be.download
eng.run_forest()
rounds = {}
for _, nA, k in log:
    rounds.setdefault(k, []).append(nA)
print("problems per k-round (summed over levels):", {k: sum(v) for k, v in sorted(rounds.items())})
print("launches:", len(log), "fits:", int(eng.counters["fits"]))
