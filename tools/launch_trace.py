"""GPU-box helper: one line per k_kmeans_restart launch of a config-C batch on ONE stream:
fits in the launch, k, device ms, largest and total D*V*k of the launch (is the launch a tail or a throughput problem?)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0)
for rep in range(2):
    eng = F.ForestEngine(be, 5, 7)
    eng.load(msas)
    be.profile = {} if rep else None
    log = []
    orig = be.call
    ptabs = {}

    def call(name, *a, **k):
        if name == "mprg_kmeans_restarts":
            log.append((ptabs[a[0]], ptabs[a[1]], a[2]))
        return orig(name, *a, **k)

    up = be.upload

    def upload(arr):
        t = up(arr)
        if isinstance(arr, np.ndarray) and arr.dtype in (np.int64, np.int32) and arr.ndim == 2:
            ptabs[be.ptr(t)] = arr.copy()
        return t

    be.call, be.upload = call, upload
    eng.run_forest()
    be.synchronize()
    be.call, be.upload = orig, up
import torch
evs = be.profile["mprg_kmeans_restarts"]
tot = 0.0
rows = []
for (pp, kp, nA), (e0, e1, _) in zip(log, evs):
    ms = e0.elapsed_time(e1)
    pt, ki = pp, kp
    d, v = pt[ki[:, 0], 1], pt[ki[:, 0], 7]
    rows.append((nA, int(ki[0, 1]), ms, int((d * v).max()), int(d.max()), int(v.max()), int((d * v).sum())))
    tot += ms
print("launches", len(rows), "total ms", round(tot, 2))
print("fits k ms maxDV maxD maxV sumDV")
for r in rows:
    print(*[round(x, 3) if isinstance(x, float) else x for x in r])
