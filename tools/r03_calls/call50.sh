#!/bin/bash
mkdir -p gpurun_out/r03_c50
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for round in 1 2 3; do for tag in nogroup dyn4 dyn2 cc4 cc2 ccsn2; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip_$tag.so
  MPRG_HIP_LIB=$lib python bench.py $o > gpurun_out/r03_c50/b_$tag.json 2> gpurun_out/r03_c50/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c50/b_$tag.json"))
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; exclusive shard: small", ks.get("mprg_kmeans_fit_small"), "general", ks.get("mprg_kmeans_fit"), "verified", b["config"]["verified"]["mismatches"])
P
done; done | tee gpurun_out/r03_c50/summary.txt
