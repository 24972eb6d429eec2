#!/bin/bash
# group loads sized per phase (largest of 4 / 2 / 1 lanes per item that fits one pass) against none; GPU parity of the new default
mkdir -p gpurun_out/r03_c49
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py -m gpu -x -q 2>&1 | tail -3
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --steps 10 --warmup 2"
for round in 1 2 3; do for tag in nogroup group; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip.so; [ $tag = nogroup ] && lib=$PWD/make_prg_amd/_lib/libmprg_hip_nogroup.so
  MPRG_HIP_LIB=$lib python bench.py $o > gpurun_out/r03_c49/b_$tag.json 2> gpurun_out/r03_c49/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c49/b_$tag.json"))
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
sw=b["config"]["single_worker"]
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; one worker", round(sw["value"]), "; exclusive shard: small", ks.get("mprg_kmeans_fit_small"), "general", ks.get("mprg_kmeans_fit"), "verified", b["config"]["verified"]["mismatches"])
P
done; done
