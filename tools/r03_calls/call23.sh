#!/bin/bash
mkdir -p gpurun_out/r03_c23
for ss in 1 0; do
MPRG_KM_SIDE_STREAMS=$ss python tools/forest_profile.py 30000 2 > gpurun_out/r03_c23/prof_side$ss.txt 2>&1
grep "^step\|pipelined" -A0 gpurun_out/r03_c23/prof_side$ss.txt | head -3; grep -A7 "pipelined" gpurun_out/r03_c23/prof_side$ss.txt | tail -6
done
python bench.py --no-cpu-baseline --no-end-to-end --no-cli-leg --steps 8 > gpurun_out/r03_c23/bench.json 2> gpurun_out/r03_c23/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r03_c23/bench.json')); print(d['value'], d['ms_per_step'], d['config']['single_worker']['value'], d['config']['verified']['mismatches'])"
