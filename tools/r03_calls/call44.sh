#!/bin/bash
mkdir -p gpurun_out/r03_c44
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for cfg in "4 1" "4 2" "2 2" "2 4" "3 2" "6 2"; do set -- $cfg
  python bench.py $o --workers $1 --streams $2 > gpurun_out/r03_c44/w$1_s$2.json 2> gpurun_out/r03_c44/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c44/w$1_s$2.json"))
print("workers $1 streams $2:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step", b["config"]["verified"]["mismatches"])
P
done
