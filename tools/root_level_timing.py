"""GPU-box helper: k_partition phase cycles of the ROOT level only (diagnostic build -DKM_PHASE_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "make_prg_amd", "_lib", "libmprg_hip_timing.so")
msas = make_batch(list(range(2048)), 16)[1]
be = HipBackend(0, lib_path=lib)
be.lib.mprg_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
eng = F.ForestEngine(be, 5, 7)
eng.load(msas)
eng.run_forest()
orig = eng._forest_level
calls = []


def level(cur):
    if not calls:
        be.synchronize(); be.lib.mprg_debug_phase_cycles(None, 1)
    out = orig(cur)
    if not calls:
        be.synchronize()
        o = (ctypes.c_ulonglong * 32)(); be.lib.mprg_debug_phase_cycles(o, 0)
        c = np.array(list(o), dtype=np.float64)[16:24]
        names = ["column flags", "serial scan", "pass A", "pass B", "merge", "copy-out", "record+atomic"]
        print("root level, per view (k cycles):", {n: round(v / len(cur["idx"]) / 1e3, 1) for n, v in zip(names, c)})
    calls.append(1)
    return out


eng._forest_level = level
eng.run_forest()
