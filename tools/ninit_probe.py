"""GPU-box helper (timing only, results are NOT the reference's): KMeans kernel time with n_init restarts side by side per
workgroup — how much of a fit's latency is shared when a workgroup carries twice the restarts?
usage: MPRG_HIP_LIB=<build with -DKM_RMAX=20> python tools/ninit_probe.py <n_init> [loci]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import make_prg_amd.engine as E
n_init = int(sys.argv[1])
E.N_INIT = n_init
import make_prg_amd.forest as F
F.N_INIT = n_init
from bench import make_batch
from make_prg_amd.backend import HipBackend

n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0)
eng = F.ForestEngine(be, 5, 7)
eng.load(msas)
eng.run_forest()
be.synchronize()
be.profile = {}
eng.counters["fits"] = 0
eng.run_forest()
be.synchronize()
s = be.profile_summary()
print("n_init", n_init, "fits", eng.counters["fits"], {k: round(v["ms"], 2) for k, v in s.items() if "kmeans" in k})
