"""GPU-box helper: hierarchical alignments (utils/synthetic.synth_rows_deep) of a size the oracle still builds in seconds, several seeds:
the HIP path (big-view kernels, wide fits, tiled tables when the level counts as big) against the oracle — PRG, tree and prg_index.
usage: deep_sweep.py [n_seeds] [rows] [cols] [big_bytes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
C = int(sys.argv[3]) if len(sys.argv) > 3 else 900
big = int(sys.argv[4]) if len(sys.argv) > 4 else 4096          # every clustering problem of some size counts as BIG (wide fits, tiled tables)

import make_prg_amd.forest as forest
from make_prg_amd.backend import HipRuntimeBackend
from make_prg_amd.utils.synthetic import synth_rows_deep
from tests import parity_common as pc

forest.KM_BIG_BYTES = big
pc.ENGINE = "forest"
be = HipRuntimeBackend(0)
t0 = time.time()
bad = 0
for seed in range(100, 100 + n_seeds):
    rows = synth_rows_deep(seed, S, C)
    text = "".join(f">d{i}\n{r}\n" for i, r in enumerate(rows))
    try:
        pc.check_vs_oracle(be, [text], 7, 7)
        print(f"seed {seed}: {S} x {C} ok", flush=True)
    except AssertionError as err:
        bad += 1
        print(f"seed {seed}: MISMATCH {str(err)[:200]}", flush=True)
print(f"{n_seeds} hierarchical alignments {S} x {C} (KM_BIG_BYTES {big}) in {time.time() - t0:.0f} s; mismatches: {bad}")
