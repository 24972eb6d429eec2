"""GPU-box helper: cProfile of one step of the ForestEngine host (single stream)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from make_prg_amd.backend import HipBackend
from make_prg_amd.forest import ForestEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
msas = make_batch(list(range(n)), 16)[1]
be = HipBackend(0)
eng = ForestEngine(be, 5, 7)
eng.load(msas)


def step():
    eng.run_forest()
    return eng.assemble_prgs(as_bytes=True)


step()
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
