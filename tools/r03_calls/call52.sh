#!/bin/bash
mkdir -p gpurun_out/r03_c52
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 20 --warmup 3"
for cfg in "3750 1" "3750 2" "3750 4" "7500 1" "7500 4" "15000 1" "15000 4"; do set -- $cfg
  python bench.py $o --batch $1 --workers $2 > gpurun_out/r03_c52/b_$1_w$2.json 2> gpurun_out/r03_c52/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c52/b_$1_w$2.json"))
print("batch $1 workers $2:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step", "side streams", b["config"].get("km_side_streams"))
P
done | tee gpurun_out/r03_c52/summary.txt
