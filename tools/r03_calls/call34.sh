#!/bin/bash
# host worker processes per GPU at the full batch; what one worker does with a small shard (device vs wall)
mkdir -p gpurun_out/r03_c34
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 10 --warmup 2"
for w in 3 4 5 6 8; do
  python bench.py $o --workers $w > gpurun_out/r03_c34/w$w.json 2> gpurun_out/r03_c34/w$w.err
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c34/w$w.json"))
print("workers $w:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step")
P
done
python tools/forest_profile.py 3750 3 > gpurun_out/r03_c34/prof_3750.txt 2>&1
grep "step 2\|device time\|kmeans\|partition\|pipelined" -A0 gpurun_out/r03_c34/prof_3750.txt | head; tail -8 gpurun_out/r03_c34/prof_3750.txt
