"""GPU-box helper: do forests on several streams of ONE process overlap on the device?  Forests only (no assembly, no waits inside):
every engine's forest is enqueued from its plan `steps` times in turn, one wait at the end.  usage: overlap_probe.py <alignments> <streams>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch, lpt_parts
from make_prg_amd.backend import make_backend
from make_prg_amd.forest import ForestEngine
n, S = int(sys.argv[1]), int(sys.argv[2])
steps = 8
parts = lpt_parts(list(range(n)), S) if S > 1 else [list(range(n))]
engs = []
for p in parts:
    be = make_backend("runtime", 0)
    e = ForestEngine(be, 5, 7)
    e.load(make_batch(p, 16)[1])
    e.run_forest(); e.run_forest()
    engs.append(e)
for e in engs: e.be.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    for e in engs:
        e.forest_enqueue()
t1 = time.perf_counter()
for e in engs: e.be.synchronize()
t2 = time.perf_counter()
print(f"{n} alignments on {S} streams: {steps} forests each enqueued in {1e3*(t1-t0):.1f} ms, done after {1e3*(t2-t0):.1f} ms -> {1e3*(t2-t0)/steps:.1f} ms per pass, {n*steps/(t2-t0):.0f} alignments/s (forests only)")
