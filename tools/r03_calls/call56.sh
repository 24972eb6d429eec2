#!/bin/bash
mkdir -p gpurun_out/r03_c56
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 20 --warmup 3"
for cfg in "3750 1 1" "3750 1 2" "3750 1 3" "7500 1 1" "7500 1 2" "30000 1 1" "30000 1 2"; do set -- $cfg
  python bench.py $o --batch $1 --workers $2 --streams $3 > gpurun_out/r03_c56/b_$1_w$2_s$3.json 2> gpurun_out/r03_c56/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c56/b_$1_w$2_s$3.json"))
print("batch $1 workers $2 streams $3:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step", b["config"]["verified"]["mismatches"])
P
done | tee gpurun_out/r03_c56/summary.txt
