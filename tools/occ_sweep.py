"""GPU-box helper: bench with alternative builds of the library (occupancy variants of k_kmeans_restart)."""
import json, os, shutil, subprocess, sys
lib = "make_prg_amd/_lib/libmprg_hip.so"
shutil.copy(lib, lib + ".orig")
try:
    for w in (4, 6, 8):
        shutil.copy(f"make_prg_amd/_lib/libmprg_hip_w{w}.so", lib)
        for st in (1, 3):
            out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--streams", str(st)],
                                 capture_output=True, text=True).stdout.strip().splitlines()
            d = json.loads(out[-1])
            print("waves", w, "streams", st, "value", d["value"], "dev_ms", d["config"]["device_ms_per_step"],
                  "restarts", d["config"]["kernels"]["mprg_kmeans_restarts"], flush=True)
finally:
    shutil.copy(lib + ".orig", lib)
