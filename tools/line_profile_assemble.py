"""GPU-box helper: per-line wall time of forest.assemble_prgs (sys.settrace on that one frame), batch from argv."""
import os
import sys
import time
import collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fname = sys.argv[2] if len(sys.argv) > 2 else "assemble_prgs"
msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0)
eng = F.ForestEngine(be, 5, 7)
eng.load(msas)
eng.run_forest(); eng.assemble_prgs(as_bytes=True)
acc = collections.Counter()
state = {"t": None, "line": None}
code_names = {fname}


def tracer(frame, event, arg):
    if frame.f_code.co_name not in code_names:
        return None

    def local(frame, event, arg):
        now = time.perf_counter()
        if state["line"] is not None:
            acc[state["line"]] += now - state["t"]
        state["t"], state["line"] = time.perf_counter(), frame.f_lineno
        if event == "return":
            state["line"] = None
        return local
    state["t"], state["line"] = time.perf_counter(), frame.f_lineno
    return local


sys.settrace(tracer)
eng.run_forest(); be.synchronize()
eng.assemble_prgs(as_bytes=True)
sys.settrace(None)
src = open(F.__file__).read().splitlines()
for line, t in acc.most_common(25):
    print(f"{t*1e3:8.1f} ms  {line:4d}  {src[line-1].strip()[:110]}")
