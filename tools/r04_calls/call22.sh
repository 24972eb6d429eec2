#!/bin/bash
# big views spread over the chip (scan / majority tiles / gap-run segments / wide row chunks / bit-array scan / wave-laid leaves):
# config D timed + kernel stats, the GPU tests that touch big views, one config C bench for regressions
out=gpurun_out/r04_c22; mkdir -p $out
export TMPDIR=/tmp
MPRG_CONFIG_D_OUT=$out/config_d_timing.json python tools/config_d_profile.py --passes 3 2>&1 | tee $out/config_d_timing.txt | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv; head -14 $f | cut -c1-150
rm -rf $out/prof
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $out/pytest_gpu.txt
python bench.py --no-shard-projection > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/r04_c22/bench.json'))
print(b['value'], b['ms_per_step'], b['config']['single_worker']['value'], b['verified']['mismatches'] if 'verified' in b else b['config'].get('verified'))
PY
