"""GPU-box helper: a HIERARCHICAL alignment (utils/synthetic.synth_rows_deep: the recursion nests down to the nesting limit, unlike
the flat generator of BASELINE config D) timed per entry point: the parity-checked `ddeep` shape (2 000 x 4 000, -N 7) by default, or
    python tools/deep_profile.py S C [N] [--passes K] [--check ROWS]
--check: the PRG must spell ROWS sampled input rows exactly once each (tests/prg_walk.py; no oracle beyond the fixtures' sizes)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from make_prg_amd.backend import make_backend
from make_prg_amd.forest import ForestEngine
from make_prg_amd.msa import MSA, Record
from make_prg_amd.utils.synthetic import synth_rows_deep

argv, args, opts = sys.argv[1:], [], {}
while argv:
    a = argv.pop(0)
    if a.startswith("--"):
        opts[a] = int(argv.pop(0))
    else:
        args.append(a)
S, C = (int(args[0]), int(args[1])) if len(args) >= 2 else (2000, 4000)
N = int(args[2]) if len(args) >= 3 else 7
passes, check = opts.get("--passes", 3), opts.get("--check", 0)
t0 = time.time()
rows = synth_rows_deep(0, S, C)
msa = MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(rows)])
print(f"generated {S} x {C} in {time.time() - t0:.1f} s", flush=True)
be = make_backend(os.environ.get("MPRG_BACKEND", "torch"), 0)
eng = ForestEngine(be, N, 7)
t0 = time.perf_counter()
eng.load([msa])
be.synchronize()
print(f"ingest {time.perf_counter() - t0:.3f} s", flush=True)
out = dict(config=f"deep: one hierarchical alignment {S} x {C}, -N {N} -L 7", passes=[])
prg = None
for p in range(passes):
    for k in eng.counters:
        eng.counters[k] = 0 if k != "arena_bytes" else eng.counters[k]
    be.profile = {} if p == 0 else None
    be.synchronize()
    t0 = time.perf_counter()
    eng.run_forest()
    t1 = time.perf_counter()
    prg = eng.assemble_prgs(as_bytes=True)[0]
    be.synchronize()
    t2 = time.perf_counter()
    import hashlib
    rec = dict(prg_sha256=hashlib.sha256(bytes(prg)).hexdigest(), forest_ms=round(1e3 * (t1 - t0), 2), assemble_ms=round(1e3 * (t2 - t1), 2), wall_ms=round(1e3 * (t2 - t0), 2), nodes=int(eng.n_nodes),
               levels=len(eng.levels), fits=int(eng.counters["fits"]), host_waits=int(eng.counters.get("syncs", 0)), calls=int(eng.counters["launches"]),
               prg_chars=len(prg), plan_misses=int(eng.counters.get("plan_misses", 0)),
               host="per-step host" if eng.counters.get("syncs", 0) > 3 * len(eng.levels) else "enqueued from the plan")
    if be.profile is not None:
        if os.environ.get("MPRG_PROFILE_ALL_LAUNCHES"):          # per launch: the wide fits of every level
            for name in ("mprg_kmeans_fit_wide", "mprg_kmeans_prepare_big", "mprg_cluster_further"):
                evs = be.profile.get(name, [])
                be.synchronize()
                print(f"  per launch {name}: " + " ".join(f"{a.elapsed_time(b):.1f}" for a, b, _ in evs))
        prof = be.profile_summary()
        rec["device_ms"] = round(sum(v["ms"] for v in prof.values()), 3)
        rec["entry_points"] = [dict(entry_point=k, ms=round(v["ms"], 3), calls=v["calls"], algorithmic_bytes=float(v["bytes"]),
                                    achieved_GBps=round(v["bytes"] / max(v["ms"], 1e-9) * 1e-6, 3) if v["bytes"] else None)
                               for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])]
        rec["speculative_levels"] = int(eng.counters.get("speculative_levels", 0))
        # the wide fits' algorithmic bytes twice: of the fits the reference's loop reaches (what `entry_points` credits) and of everything
        # the all-rounds-at-once launches fitted (work the reference never does, done on CUs that had none)
        rec["kmeans_bytes_reference_fits"] = float(eng.counters.get("kmeans_bytes", 0.0))
        rec["kmeans_bytes_launched_at_once"] = float(eng.counters.get("kmeans_bytes_launched_at_once", 0.0))
        be.profile = None
    out["passes"].append(rec)
    print(json.dumps({k: v for k, v in rec.items() if k != "entry_points"}), flush=True)
    for e in rec.get("entry_points", [])[:16]:
        print(f"   {e['entry_point']:34s} {e['ms']:10.3f} ms {e['calls']:5d} calls" + (f"  {e['achieved_GBps']:9.1f} GB/s algorithmic" if e["achieved_GBps"] else ""))
if check:
    # (a nested PRG may spell a row along more than one path: the parity-checked ddeep PRG — identical to the real reference's —
    #  does so for 1 of 219 sampled rows; "at least one path" is the property here, "exactly one" holds for the flat config D)
    from tests.prg_walk import parse_prg, _prepare, spellings
    t0 = time.time()
    tree = parse_prg(bytes(prg).decode())
    _prepare(tree)
    distinct = list(dict.fromkeys(r.decode().replace("-", "") for r in rows))
    distinct = distinct[::max(1, len(distinct) // check)]
    paths = [spellings(tree, r) for r in distinct]
    assert min(paths) >= 1, f"{sum(p == 0 for p in paths)} of {len(paths)} sampled rows are not spelt by the PRG"
    print(f"PRG: markers nest, {len(paths)} sampled distinct rows are spelt ({sum(p > 1 for p in paths)} of them along more than one path) "
          f"({time.time() - t0:.1f} s)")
dst = os.environ.get("MPRG_DEEP_OUT")
if dst:
    json.dump(out, open(dst, "w"), indent=1)
