#!/bin/bash
# levels without host waits (mprg_forest_level): GPU parity, then one worker at 3 750 / 30 000 alignments per step, speculative on / off
out=gpurun_out/r04_c11; mkdir -p $out
python -m pytest tests/test_gpu_speculative.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 8 --warmup 2"
run() {  # tag, batch, workers, env...
  tag=$1; B=$2; W=$3; shift 3
  env "$@" python bench.py $o --batch $B --workers $W > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
c=b["config"]
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; waits/step", c["host_waits_per_step"], "launches/step", c["launches_per_step"], "verified", c["verified"]["mismatches"])
P
}
for B in 3750 30000; do for W in 1 4; do
run spec_${B}_w$W $B $W MPRG_SPECULATIVE=1
run exact_${B}_w$W $B $W MPRG_SPECULATIVE=0
done; done
