"""Small GPU-box helper: run bench.py under a few environment / flag variants and print one line each."""
import json
import os
import subprocess
import sys

variants = [
    ("km64", {"MPRG_KM_THREADS": "64"}, ["--streams", "1"]),
    ("km128", {"MPRG_KM_THREADS": "128"}, ["--streams", "1"]),
    ("km256", {"MPRG_KM_THREADS": "256"}, ["--streams", "1"]),
    ("km64x8", {"MPRG_KM_THREADS": "64"}, ["--streams", "8"]),
    ("km128x8", {"MPRG_KM_THREADS": "128"}, ["--streams", "8"]),
    ("km256x8", {"MPRG_KM_THREADS": "256"}, ["--streams", "8"]),
    ("s1", {"MPRG_KM_THREADS": "128"}, ["--streams", "1"]),
    ("s2", {"MPRG_KM_THREADS": "128"}, ["--streams", "2"]),
    ("s3", {"MPRG_KM_THREADS": "128"}, ["--streams", "3"]),
    ("s4", {"MPRG_KM_THREADS": "128"}, ["--streams", "4"]),
    ("s2b4k", {"MPRG_KM_THREADS": "128"}, ["--streams", "2", "--batch", "4096"]),
    ("s1b4k", {"MPRG_KM_THREADS": "128"}, ["--streams", "1", "--batch", "4096"]),
    ("s2k64", {"MPRG_KM_THREADS": "64"}, ["--streams", "2"]),
    ("t2", {}, ["--streams", "2"]),
    ("t3", {}, ["--streams", "3"]),
    ("t4", {}, ["--streams", "4"]),
    ("t6", {}, ["--streams", "6"]),
    ("t4b4k", {}, ["--streams", "4", "--batch", "4096"]),
    ("t4b8k", {}, ["--streams", "4", "--batch", "8192"]),
    ("t6b8k", {}, ["--streams", "6", "--batch", "8192"]),
    ("t8b8k", {}, ["--streams", "8", "--batch", "8192"]),
    ("t8b16k", {}, ["--streams", "8", "--batch", "16384"]),
    ("t3b8k", {}, ["--streams", "3", "--batch", "8192"]),
    ("t3b1k", {}, ["--streams", "3", "--batch", "1024"]),
    ("s1k256", {"MPRG_KM_THREADS": "256"}, ["--streams", "1"]),
    ("s1k512", {"MPRG_KM_THREADS": "512"}, ["--streams", "1"]),
    ("s1k1024", {"MPRG_KM_THREADS": "1024"}, ["--streams", "1"]),
    ("s2k256", {"MPRG_KM_THREADS": "256"}, ["--streams", "2"]),
    ("s3k256", {"MPRG_KM_THREADS": "256"}, ["--streams", "3"]),
    ("s3k512", {"MPRG_KM_THREADS": "512"}, ["--streams", "3"]),
]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1:]]
for name, env, flags in variants:
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "2", "--warmup", "1"] + flags,
                         env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip().splitlines()
    try:
        d = json.loads(out[-1])
        k = d["config"]["kernels"]
        print(name, "value", d["value"], "ms/step", d["ms_per_step"], "dev_ms", d["config"]["device_ms_per_step"],
              "restarts", k["mprg_kmeans_restarts"], flush=True)
    except Exception as e:  # noqa
        print(name, "FAILED", e, out[-3:], flush=True)
