#!/bin/bash
mkdir -p gpurun_out/r03_c40
python bench.py --no-cpu-baseline --no-cli-leg --no-single-worker-leg --steps 5 --warmup 2 > gpurun_out/r03_c40/b.json 2> gpurun_out/r03_c40/b.err
tail -3 gpurun_out/r03_c40/b.err
python - <<'P'
import json
b=json.load(open("gpurun_out/r03_c40/b.json"))
print(round(b["value"]), b["config"]["end_to_end"])
P
