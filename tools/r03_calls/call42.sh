#!/bin/bash
mkdir -p gpurun_out/r03_c42
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for cfg in "30000 4" "30000 1" "3750 1" "3750 4"; do set -- $cfg; for s in 0 1 0 1; do
  MPRG_KM_SIDE_STREAMS=$s python bench.py $o --batch $1 --workers $2 > gpurun_out/r03_c42/b_$1_w$2_side$s.json 2> gpurun_out/r03_c42/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c42/b_$1_w$2_side$s.json"))
print("batch $1 workers $2 side $s:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step", b["config"]["verified"]["mismatches"])
P
done; done
