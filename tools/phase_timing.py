"""GPU-box helper: share of k_kmeans_restart's workgroup time per phase (diagnostic build -DKM_PHASE_TIMING,
make_prg_amd/_lib/libmprg_hip_timing.so: shader-clock cycles of thread 0 between barriers, summed over all fits).
Build it first (in the container: hipcc cross-compiles; the .so travels with gpurun):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DKM_PHASE_TIMING \
        -Iinclude make_prg_amd/csrc/mprg_api.hip -o make_prg_amd/_lib/libmprg_hip_timing.so
The timers perturb kernels that run many short workgroups (one atomic per workgroup and mark): trust them for the KMeans
fit, not for the small-view partition.
The tool runs the PER-ROUND loop (MPRG_KLOOP=rounds unless the caller sets it): the fused-loop kernels of the diagnostic build fault on the
device (round 5: a memory access fault at a low address in k_cluster_loop_small; the product build of the same sources passes every GPU
test, and the CPU emulation of the diagnostic build passes the forest cases) — not looked into further, the per-round kernels carry the marks."""
import ctypes
import os
import sys
os.environ.setdefault("MPRG_KLOOP", "rounds")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] not in ("deep", "flat") else 2048
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "make_prg_amd", "_lib", "libmprg_hip_timing.so")
if len(sys.argv) > 1 and sys.argv[1] == "deep":          # python tools/phase_timing.py deep S C: one hierarchical alignment (-N 7)
    from make_prg_amd.msa import MSA, Record
    from make_prg_amd.utils.synthetic import synth_rows_deep
    msas = [MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(synth_rows_deep(0, int(sys.argv[2]), int(sys.argv[3])))])]
elif len(sys.argv) > 1 and sys.argv[1] == "flat":          # python tools/phase_timing.py flat S C: BASELINE config D's generator (-N 7)
    from make_prg_amd.msa import MSA, Record
    from make_prg_amd.utils.synthetic import synth_rows
    msas = [MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(synth_rows(0, int(sys.argv[2]), int(sys.argv[3]), 8))])]
else:
    msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0, lib_path=lib)
print('loop:', F.KLOOP)
eng = F.ForestEngine(be, 7 if len(msas) == 1 else 5, 7)
eng.load(msas)
eng.run_forest()
be.synchronize()
out = (ctypes.c_ulonglong * 32)()
be.lib.mprg_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
be.lib.mprg_debug_phase_cycles(None, 1)
eng.run_forest()
be.synchronize()
be.lib.mprg_debug_phase_cycles(out, 0)
c = np.array(list(out), dtype=np.float64)
names = ["k-means++ pick", "k-means++ score + first centres", "centre-centre distances", "sample-centre distances",
         "init bounds / E-step", "M-step", "relocation (slow path)", "shifts + norms", "bounds + stop test", "inertia",
         "k-means++ first centre (pick of round 1)", "first centres copied + counts staged in LDS",
         "fused loop: best restart + predict", "fused loop: cluster_further"]
for nm, v in zip(names, c):
    print(f"{100 * v / c[:16].sum():6.1f} %  {nm}   ({v / max(eng.counters['fits'], 1):.0f} cycles per fit)")
pn = ["gap runs (small views) / column flags (big views)", "serial scan", "pass A (N rows)", "pass B (row comparison)", "merge", "packed copy",
      "record", "small views: view + row indices", "small views: cells into LDS", "small views: masks + column flags"]
for nm, v in zip(pn, c[16:]):
    print(f"{100 * v / max(c[16:].sum(), 1):6.1f} %  k_partition: {nm}   ({v / 1e6:.1f} Mcycles)")
print("fits", eng.counters["fits"], "cycles/fit", c[:16].sum() / max(eng.counters["fits"], 1))
