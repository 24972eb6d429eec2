#!/bin/bash
# the library built with -mllvm -disable-machine-licm (KMeans kernels: 6 instead of 18 spilled VGPRs) against the default build,
# per-round launches and fused loop
out=gpurun_out/r04_c09; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 6 --warmup 2"
run() {  # tag, batch, workers, env...
  tag=$1; B=$2; W=$3; shift 3
  env "$@" python bench.py $o --batch $B --workers $W > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
x=b["roofline"]["exclusive_pass"]
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; excl wall", x["wall_ms"], "device", x["device_ms"], "verified", b["config"]["verified"]["mismatches"], ks)
P
}
L=$PWD/make_prg_amd/_lib
for rep in 1 2; do
run rounds_default_w4 30000 4 MPRG_KLOOP=rounds
run rounds_nomlicm_w4 30000 4 MPRG_KLOOP=rounds MPRG_HIP_LIB=$L/libmprg_hip_nomlicm.so
run fused_default_w4 30000 4 MPRG_KLOOP=fused
run fused_nomlicm_w4 30000 4 MPRG_KLOOP=fused MPRG_HIP_LIB=$L/libmprg_hip_nomlicm.so
done
