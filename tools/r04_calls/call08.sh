#!/bin/bash
# fused loop with the round's fit as a function of its own (no values hoisted over the loop body)
out=gpurun_out/r04_c08; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 6 --warmup 2"
run() {  # tag, batch, workers, env...
  tag=$1; B=$2; W=$3; shift 3
  env "$@" python bench.py $o --batch $B --workers $W > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
x=b["roofline"]["exclusive_pass"]
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; excl wall", x["wall_ms"], "device", x["device_ms"], "launches", x["launches"], "waits", x["host_waits"], "verified", b["config"]["verified"]["mismatches"], ks)
P
}
for W in 1 4; do
run fused_30000_w$W 30000 $W MPRG_KLOOP=fused
done
run fused_3750_w1 3750 1 MPRG_KLOOP=fused
