#!/bin/bash
# round 6, call 29: how the per-round LDS fits' time follows the number of workgroups a CU holds — the same kernel and layout, launched with
# more dynamic LDS than it uses (a test build: MPRG_KML_PAD_PCT, workgroups per CU = 100 / (100 + pad) of the class's)
out=gpurun_out/r06_c29; mkdir -p $out
export TMPDIR=/tmp
export MPRG_HIP_LIB=$GRAFT_REPO_ROOT/make_prg_amd/_lib/libmprg_hip_pad.so
for pad in 0 15 34 60 100; do
  MPRG_KML_PAD_PCT=$pad timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_pad$pad.txt 2>&1
  echo "== pad $pad %"; grep "device time\|  mprg_kmeans_fit_lds" $out/forest_7500_pad$pad.txt
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for pad in 0 34 100 0; do
  MPRG_KML_PAD_PCT=$pad timeout 600 python bench.py $quick > $out/bench_pad${pad}_$RANDOM.json 2> $out/bench_err.txt
  g=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$g').read().strip().splitlines()[-1]); print('30000 pad $pad:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done
