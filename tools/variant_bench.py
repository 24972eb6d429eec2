"""GPU-box helper: device time of k_kmeans_restart with a diagnostic build (w4 = k-means++ only) vs the real one."""
import json, shutil, subprocess, sys
lib = "make_prg_amd/_lib/libmprg_hip.so"
shutil.copy(lib, lib + ".orig")
try:
    for tag, src in (("kmeans++ only (results wrong, timing build)", "make_prg_amd/_lib/libmprg_hip_w4.so"), ("full", lib + ".orig")):
        shutil.copy(src, lib)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "2", "--warmup", "1", "--streams", "1", "--batch", "2048"],
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(tag, "restarts", d["config"]["kernels"]["mprg_kmeans_restarts"], "fits", d["config"]["kmeans_fits_per_step"], flush=True)
        except Exception as e:
            print(tag, "failed", e, out.stderr[-500:])
finally:
    shutil.copy(lib + ".orig", lib)
