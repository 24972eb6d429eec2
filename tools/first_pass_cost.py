"""GPU-box helper: one engine, n alignments, a forest enqueued from its OWN plan beside one enqueued from ANOTHER batch's totals (first pass):
host time of forest_enqueue, wall until forest_finish returns, launches and capacities — where a first pass loses against a planned one.
usage: first_pass_cost.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import make_backend
from make_prg_amd.forest import ForestEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3750
_, msas = make_batch(list(range(n)), 16)
_, calib = make_batch(list(range(2_000_000, 2_000_000 + n)), 16)
be = make_backend("runtime", 0)
ce = ForestEngine(be, 5, 7); ce.load(calib); ce.run_forest(); donor = ce.plan_export()
eng = ForestEngine(be, 5, 7); eng.load(msas); eng.run_forest(); eng.assemble_prgs(as_bytes=True)
own = eng._plan
for mode in ("planned", "first", "planned", "first", "planned", "first"):
    be.synchronize()
    if mode == "first":
        eng._plan = None; eng.plan_donor = donor
    else:
        eng._plan = own
    c0 = dict(eng.counters)
    t0 = time.perf_counter(); eng.forest_enqueue(); t1 = time.perf_counter()
    eng.forest_finish(); be.synchronize(); t2 = time.perf_counter()
    own = eng._plan if mode == "planned" else own
    print(f"{mode:8s}: enqueue {1e3*(t1-t0):6.2f} ms (host), forest done after {1e3*(t2-t0):6.2f} ms; levels {len(eng.levels)}; launches {eng.counters['launches'] - c0['launches']}; "
          f"misses {eng.counters.get('plan_misses', 0) - c0.get('plan_misses', 0)}", flush=True)
caps_p, caps_f = eng._caps_own(own), eng._caps_predicted(donor)
print("levels planned / first:", len(caps_p["levels"]), len(caps_f["levels"]))
