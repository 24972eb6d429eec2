#!/bin/bash
mkdir -p gpurun_out/r03_c51
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for round in 1 2 3; do for tag in nopredict predict; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip.so; [ $tag = nopredict ] && lib=$PWD/make_prg_amd/_lib/libmprg_hip_nopredict.so
  MPRG_HIP_LIB=$lib python bench.py $o > gpurun_out/r03_c51/b_$tag.json 2> gpurun_out/r03_c51/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c51/b_$tag.json"))
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; exclusive shard: small", ks.get("mprg_kmeans_fit_small"), "general", ks.get("mprg_kmeans_fit"), "verified", b["config"]["verified"]["mismatches"])
P
done; done | tee gpurun_out/r03_c51/summary.txt
