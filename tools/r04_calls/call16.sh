#!/bin/bash
# compact restart regions (kcap = k): six-form KMeans parity, then both loop forms at 30 000 / 3 750
out=gpurun_out/r04_c16; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_ddeep.py -m gpu -x -q 2>&1 | tail -2
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 8 --warmup 2"
run() {  # tag, batch, workers, streams, env...
  tag=$1; B=$2; W=$3; S=$4; shift 4
  env "$@" python bench.py $o --batch $B --workers $W --streams $S > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
c=b["config"]
x=b["roofline"]["exclusive_pass"]
ks={k["entry_point"]:k["ms"] for k in b["roofline"]["kernels"][:3]}
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; excl device", x["device_ms"], "verified", c["verified"]["mismatches"], ks)
P
}
for rep in 1 2; do
run fused_w4 30000 4 1 MPRG_KLOOP=fused
run rounds_w4 30000 4 1 MPRG_KLOOP=rounds
done
run fused_w1 30000 1 1 MPRG_KLOOP=fused
run fused_3750_w1 3750 1 1 MPRG_KLOOP=fused
