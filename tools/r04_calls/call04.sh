#!/bin/bash
# fused loop after giving the k = 7..10 small variant three waves' registers; does the scratch size (336 B per lane) throttle residency?
out=gpurun_out/r04_c04; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 6 --warmup 2"
run() {  # tag, env...
  tag=$1; shift
  env "$@" python bench.py $o --batch 30000 --workers $W > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
x=b["roofline"]["exclusive_pass"]
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; excl wall", x["wall_ms"], "device", x["device_ms"], "launches", x["launches"], "waits", x["host_waits"], "verified", b["config"]["verified"]["mismatches"], ks)
P
}
for W in 1 4; do
run fused_w$W MPRG_KLOOP=fused
run fused_scratch_w$W MPRG_KLOOP=fused HSA_SCRATCH_SINGLE_LIMIT=1000000000
run rounds_w$W MPRG_KLOOP=rounds
done
MPRG_BACKEND=runtime python tools/forest_profile.py 7500 4 > $out/forest_7500_fused.txt 2>&1
grep -E "^step 3|device time|cluster_loop|cluster_further" $out/forest_7500_fused.txt | cut -c1-150
