import sys, numpy as np
sys.path.insert(0, '/root/repo')
from make_prg_amd.backend import make_backend
import make_prg_amd.forest as F
from make_prg_amd.msa import MSA, Record
from make_prg_amd.utils.synthetic import synth_rows
S, C = int(sys.argv[1]), int(sys.argv[2])
rows = synth_rows(0, S, C, 8)
msa = MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(rows)])
be = make_backend(sys.argv[3] if len(sys.argv) > 3 else "runtime", 0)
eng = F.ForestEngine(be, 7, 7)
eng.load([msa])
eng.run_forest()
orig = eng._forest_speculative_finish
def fin(plan, d_ds, n_words, reps):
    ds = be.download(d_ds, np.int64, n_words)
    print("ds global", ds[:8].tolist())
    for li in range(len(plan["levels"])):
        blk = ds[16 + li*576: 16+(li+1)*576].reshape(6, 96)
        for s_ in range(6):
            a, b = blk[s_][:16].tolist(), plan["levels"][li][s_][:16].tolist()
            if a != b: print("level", li, "step", s_, "dev", a, "plan", b)
    return orig(plan, d_ds, n_words, reps)
eng._forest_speculative_finish = fin
eng.run_forest()
print("misses", eng.counters.get("plan_misses"))
