"""Config D (single deep MSA) on the GPU: timing + size-independent checks (no oracle at full size).
usage: python tools/deep_config.py S C [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from make_prg_amd.backend import HipBackend
from make_prg_amd.forest import ForestEngine
from make_prg_amd.msa import MSA, Record
from make_prg_amd.utils.synthetic import synth_rows

S, C = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 7
t0 = time.time()
rows = synth_rows(0, S, C, 8)
msa = MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(rows)])
print(f"generated {S}x{C} in {time.time() - t0:.1f}s", flush=True)
be = HipBackend(0)
eng = ForestEngine(be, N, 7)
t0 = time.time()
eng.load([msa])
t1 = time.time()
be.profile = {}
eng.run_forest()
prgs = eng.assemble_prgs()
be.synchronize()
t2 = time.time()
prg = prgs[0]
print(f"load {t1 - t0:.2f}s  build+emit {t2 - t1:.2f}s  nodes {eng.T.n}  fits {int(eng.counters['fits'])}  prg chars {len(prg)}")
for k, v in sorted(be.profile_summary().items(), key=lambda kv: -kv[1]["ms"]):
    gb = v["bytes"] / max(v["ms"], 1e-9) * 1e-6
    print(f"  {k:24s} {v['ms']:10.2f} ms  calls {v['calls']:3d}  alg GB/s {gb:8.1f}")
# size-independent properties: the PRG spells every distinct ungapped input sequence when it is one multi-allele site,
# the binary encoding round-trips its site markers, and every allele is non-empty
units = prg.split()
alleles = [u for u in units if not u.isdigit()]
assert all(alleles), "empty allele"
markers = [int(u) for u in units if u.isdigit()]
assert all(m >= 5 for m in markers)
distinct = {r.replace(b"-", b"").decode() for r in rows}
if eng.T.n == 2 and markers:          # root + one leaf listing every distinct sequence
    assert set(alleles) == distinct and len(alleles) == len(distinct)
    print("property ok: alleles == distinct ungapped rows", len(distinct))
print("ok")
