#!/bin/bash
# centre-centre distances with a quad of lanes per pair (16-byte pieces of a row's 64-byte blocks, passed round by DPP): A/B
mkdir -p gpurun_out/r03_c46
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for tag in base quad base quad; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip.so; [ $tag = quad ] && lib=$PWD/make_prg_amd/_lib/libmprg_hip_quad.so
  MPRG_HIP_LIB=$lib python bench.py $o > gpurun_out/r03_c46/b_$tag.json 2> gpurun_out/r03_c46/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c46/b_$tag.json"))
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; exclusive shard: small", ks.get("mprg_kmeans_fit_small"), "general", ks.get("mprg_kmeans_fit"), "verified", b["config"]["verified"]["mismatches"])
P
done
for tag in base quad; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip.so; [ $tag = quad ] && lib=$PWD/make_prg_amd/_lib/libmprg_hip_quad.so
  MPRG_HIP_LIB=$lib MPRG_BACKEND=runtime python tools/forest_profile.py 30000 2 > gpurun_out/r03_c46/prof_$tag.txt 2>&1
  echo "== $tag"; grep "kmeans_fit\|device time" gpurun_out/r03_c46/prof_$tag.txt
done
