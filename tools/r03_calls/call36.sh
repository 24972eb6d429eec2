#!/bin/bash
mkdir -p gpurun_out/r03_c36
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --steps 10 --warmup 2"
for b in runtime torch runtime torch; do
  MPRG_BACKEND=$b python bench.py $o > gpurun_out/r03_c36/$b.json 2> gpurun_out/r03_c36/$b.err
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c36/$b.json"))
sw=b["config"]["single_worker"]
print("$b:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; single worker", round(sw["value"]), sw["fraction_of_value"], "excl", sw["exclusive_pass"]["wall_over_device"], "verified", b["config"]["verified"]["mismatches"])
P
done
