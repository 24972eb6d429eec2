"""GPU-box helper: statistics of the KMeans fits of a config-C batch — how many fits, and where the work (8 D V per Elkan
iteration and restart seeding) sits by problem size and k."""
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
rows = []
orig = F.ForestEngine._kmeans_round


def round_(self, active, k, D, V, *a):
    act, st, info = orig(self, active, k, D, V, *a)
    rows.append(np.stack([D[act], V[act], np.full(len(act), k), info[:, 4], info[:, 1]], 1))
    return act, st, info


F.ForestEngine._kmeans_round = round_
eng.run_forest()
r = np.concatenate(rows)
D, V, k, it = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
work = D * V * (it + 10)
print("fits", len(r), "launch rounds", len(rows), "fits per locus", round(len(r) / n, 1))
print("mean D %.1f  V %.1f  k %.2f  Elkan iterations per fit (10 restarts) %.1f" % (D.mean(), V.mean(), k.mean(), it.mean()))
for name, x, edges in (("D", D, [0, 8, 16, 32, 64, 128, 1 << 30]), ("V", V, [0, 32, 64, 128, 256, 512, 1024, 1 << 30]),
                       ("k", k, [2, 3, 4, 5, 6, 8, 11]), ("D*V doubles", D * V, [0, 512, 2048, 8192, 32768, 1 << 40])):
    print(name)
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (x >= lo) & (x < hi)
        print("  [%d, %d): fits %5.1f %%  work %5.1f %%" % (lo, min(hi, 99999999), 100 * m.mean(), 100 * work[m].sum() / work.sum()))
print("fits per launch round:", [len(x) for x in rows])
