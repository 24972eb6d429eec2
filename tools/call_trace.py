"""GPU-box helper: one line per launch of a chosen entry point on ONE stream: integer arguments and device ms.
usage: call_trace.py <entry point> [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0)
eng = F.ForestEngine(be, 5, 7)
eng.load(msas)
eng.run_forest(); eng.assemble_prgs(as_bytes=True)
be.profile = {}
log = []
orig = be.call


def call(nm, *a, **k):
    if nm == name:
        log.append([x for x in a if isinstance(x, int) and abs(x) < (1 << 31)])
    return orig(nm, *a, **k)


be.call = call
eng.run_forest(); eng.assemble_prgs(as_bytes=True)
be.synchronize()
tot = 0.0
for ints, (e0, e1, w) in zip(log, be.profile[name]):
    ms = e0.elapsed_time(e1)
    tot += ms
    print(f"{ms:9.3f} ms  ints={ints}  work={w:.3g}")
print("total", round(tot, 2), "ms in", len(log), "launches")
