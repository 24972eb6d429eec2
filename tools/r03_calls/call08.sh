#!/bin/bash
mkdir -p gpurun_out/r03_c08
for cfg in "1 1" "1 2" "1 4" "2 2"; do set -- $cfg
python bench.py --no-cpu-baseline --no-end-to-end --workers $1 --streams $2 --steps 6 --warmup 2 > gpurun_out/r03_c08/bench_w$1s$2.json 2> gpurun_out/r03_c08/bench_w$1s$2.err
done
