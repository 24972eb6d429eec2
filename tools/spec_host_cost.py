"""GPU-box helper: host time of a forest enqueued from a plan (forest_enqueue), of forest_finish and of assemble_prgs, one engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from make_prg_amd.backend import make_backend
from make_prg_amd.forest import ForestEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 940
texts, msas = make_batch(list(range(n)), 16)
be = make_backend("runtime", 0)
eng = ForestEngine(be, 5, 7)
eng.load(msas)
eng.run_forest(); eng.assemble_prgs(as_bytes=True)
for s in range(6):
    be.synchronize()
    t0 = time.perf_counter(); eng.forest_enqueue(); t1 = time.perf_counter()
    eng.forest_finish(); t2 = time.perf_counter()
    fin = eng.assemble_prgs(as_bytes=True, lazy=True); t3 = time.perf_counter()
    fin(); t4 = time.perf_counter()
    print(f"step {s}: enqueue {1e3*(t1-t0):.2f} ms, finish (wait) {1e3*(t2-t1):.2f} ms, assemble {1e3*(t3-t2):.2f} ms, collect {1e3*(t4-t3):.2f} ms; levels {len(eng.levels)}")
