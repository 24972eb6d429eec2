"""GPU-box helper: where one step of the device-resident forest spends its wall time (one process, one stream).
    python tools/forest_profile.py [batch] [steps]
Prints per step: wall of run_forest / assemble_prgs, time inside downloads (waits for the device + copies), number of
waits, and — last step, HIP events on — the device time per entry point."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import HipBackend, HipRuntimeBackend
from make_prg_amd.forest import ForestEngine

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
texts, msas = make_batch(list(range(batch)), 16)
be = HipRuntimeBackend(0) if os.environ.get('MPRG_BACKEND') == 'runtime' else HipBackend(0)
eng = ForestEngine(be, 5, 7)
eng.load(msas)
be.synchronize()
dl = dict(t=0.0, n=0, bytes=0)
orig = be.download
def timed_download(buf, dtype, count):
    t0 = time.perf_counter()
    out = orig(buf, dtype, count)
    dl["t"] += time.perf_counter() - t0; dl["n"] += 1; dl["bytes"] += out.nbytes
    return out
be.download = timed_download
for s in range(steps):
    last = s == steps - 1
    be.profile = {} if last else None
    for k in eng.counters:
        if k != "arena_bytes": eng.counters[k] = 0
    dl.update(t=0.0, n=0, bytes=0)
    t0 = time.perf_counter()
    eng.run_forest()
    t1 = time.perf_counter(); d1 = dict(dl)
    prgs = eng.assemble_prgs(as_bytes=True)
    be.synchronize()
    t2 = time.perf_counter()
    print(f"step {s}: run_forest {1e3*(t1-t0):.1f} ms (downloads {1e3*d1['t']:.1f} ms in {d1['n']} waits), assemble {1e3*(t2-t1):.1f} ms "
          f"(downloads {1e3*(dl['t']-d1['t']):.1f} ms, {(dl['bytes']-d1['bytes'])/1e6:.1f} MB), nodes {eng.n_nodes}, levels {len(eng.levels)}, "
          f"calls {eng.counters['launches']}", flush=True)
be.synchronize()
for name, evs in be.profile.items():          # per launch: the clustering loop / KMeans launches of every level
    if "cluster_loop" in name or "kmeans_fit" in name or (os.environ.get("MPRG_PROFILE_ALL_LAUNCHES") and name in
                                                            ("mprg_partition", "mprg_ungap_dedupe", "mprg_cluster_further", "mprg_kmeans_prepare", "mprg_column_masks")):
        print(f"  per launch {name}: " + " ".join(f"{a.elapsed_time(b):.2f}" for a, b, _ in evs))
prof = be.profile_summary()
tot = sum(v["ms"] for v in prof.values())
print(f"device time in entry points: {tot:.2f} ms")
for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    gb = v["bytes"] / max(v["ms"], 1e-9) * 1e-6 if v["bytes"] else 0
    print(f"  {name:34s} {v['ms']:9.3f} ms {v['calls']:5d} calls  {gb:8.1f} GB/s")

# ---- pipelined steps (what bench.py times): the text of step s is collected after step s+1 has been enqueued
be.profile = None
pending = None
print("pipelined: run_forest / assemble(lazy) / collect of the previous step, ms")
for s in range(steps + 3):
    t0 = time.perf_counter()
    eng.run_forest()
    t1 = time.perf_counter()
    fin = eng.assemble_prgs(as_bytes=True, lazy=True)
    t2 = time.perf_counter()
    if pending is not None:
        prgs = pending()
        n_ok = sum(p is not None for p in prgs); chars = sum(len(p) for p in prgs if p is not None)
    t3 = time.perf_counter()
    pending = fin
    print(f"  step {s}: {1e3*(t1-t0):7.1f} {1e3*(t2-t1):7.1f} {1e3*(t3-t2):7.1f}   total {1e3*(t3-t0):7.1f}", flush=True)
