#!/bin/bash
# group-fetched row walks (KM_QG lanes per item in centre-centre / shifts / norms; KM_QG_INERTIA in the inertia phase): A/B/...
mkdir -p gpurun_out/r03_c47
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for round in 1 2; do for tag in qg0 qg2 qg4 qg4i2 qg4i4; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip_$tag.so
  MPRG_HIP_LIB=$lib python bench.py $o > gpurun_out/r03_c47/b_$tag.json 2> gpurun_out/r03_c47/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c47/b_$tag.json"))
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; exclusive shard: small", ks.get("mprg_kmeans_fit_small"), "general", ks.get("mprg_kmeans_fit"), "verified", b["config"]["verified"]["mismatches"])
P
done; done
