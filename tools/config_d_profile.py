"""GPU-box helper: BASELINE config D (ONE alignment, 10 000 x 20 000, -N 7 -L 7) timed — wall and device time of the full build,
per entry point (HIP events on the launch stream, exclusive), for the per-step host and for a forest enqueued from a plan.
    python tools/config_d_profile.py [rows cols nesting] [--passes K]
Under rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE in passes of their own) use --passes 1 --no-events."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from make_prg_amd.backend import make_backend
from make_prg_amd.forest import ForestEngine
from make_prg_amd.msa import MSA, Record
from make_prg_amd.utils.synthetic import synth_rows

args = [a for a in sys.argv[1:] if not a.startswith("--")]
S, C, N = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (10_000, 20_000, 7)
passes = int(sys.argv[sys.argv.index("--passes") + 1]) if "--passes" in sys.argv else 3
events = "--no-events" not in sys.argv
t0 = time.time()
rows = synth_rows(0, S, C, 8)
msa = MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(rows)])
t_gen = time.time() - t0
be = make_backend(os.environ.get("MPRG_BACKEND", "torch"), 0)
eng = ForestEngine(be, N, 7)
t0 = time.perf_counter()
eng.load([msa])
be.synchronize()
t_load = time.perf_counter() - t0
out = dict(config=f"D: one alignment {S} x {C}, -N {N} -L 7", generate_s=round(t_gen, 2), ingest_s=round(t_load, 3), passes=[])
for p in range(passes):
    for k in eng.counters:
        eng.counters[k] = 0 if k != "arena_bytes" else eng.counters[k]
    be.profile = {} if (events and p == 0) else None          # pass 0: per entry point events (the per-step host); later passes from the plan
    be.synchronize()
    t0 = time.perf_counter()
    eng.run_forest()
    t1 = time.perf_counter()
    prg = eng.assemble_prgs(as_bytes=True)[0]          # (the text as bytes in pinned memory: what the command line's writers take)
    be.synchronize()
    t2 = time.perf_counter()
    rec = dict(forest_ms=round(1e3 * (t1 - t0), 2), assemble_ms=round(1e3 * (t2 - t1), 2), wall_ms=round(1e3 * (t2 - t0), 2), nodes=int(eng.n_nodes),
               levels=len(eng.levels), fits=int(eng.counters["fits"]), host_waits=int(eng.counters.get("syncs", 0)), calls=int(eng.counters["launches"]),
               prg_chars=len(prg), plan_misses=int(eng.counters.get("plan_misses", 0)), host="per-step host" if eng.counters.get("syncs", 0) > 6 else "enqueued from the plan",
               cells_all=eng.counters["cells_all"], cells_clustered=eng.counters["cells_clustered"], kmeans_bytes=eng.counters["kmeans_bytes"])
    if be.profile is not None:
        prof = be.profile_summary()
        rec["device_ms"] = round(sum(v["ms"] for v in prof.values()), 3)
        rec["entry_points"] = [dict(entry_point=k, ms=round(v["ms"], 3), calls=v["calls"], algorithmic_bytes=v["bytes"],
                                    GBps=round(v["bytes"] / max(v["ms"], 1e-9) * 1e-6, 1) if v["bytes"] else None,
                                    frac_of_8TBps=round(v["bytes"] / max(v["ms"], 1e-9) * 1e-6 / 8000.0, 4) if v["bytes"] else None)
                               for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])]
        be.profile = None
    out["passes"].append(rec)
    print(json.dumps({k: v for k, v in rec.items() if k != "entry_points"}), flush=True)
    for e in rec.get("entry_points", [])[:14]:
        print(f"   {e['entry_point']:34s} {e['ms']:10.3f} ms {e['calls']:4d} calls  {e['GBps'] or 0:9.1f} GB/s  frac {e['frac_of_8TBps'] or 0}")
dst = os.environ.get("MPRG_CONFIG_D_OUT")
if dst:
    json.dump(out, open(dst, "w"), indent=1)
