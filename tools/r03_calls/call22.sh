#!/bin/bash
mkdir -p gpurun_out/r03_c22
for cfg in "1 2" "1 3" "2 2" "4 1" "6 1"; do set -- $cfg
python bench.py --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --workers $1 --streams $2 --steps 6 --warmup 2 > gpurun_out/r03_c22/bench_w$1s$2.json 2> gpurun_out/r03_c22/bench_w$1s$2.err
python -c "
import json; d=json.load(open('gpurun_out/r03_c22/bench_w$1s$2.json')); print('w$1 s$2', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done
