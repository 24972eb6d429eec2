#!/bin/bash
# round 6, call 31: more threads per fit for the LDS classes of 3 / 2 / 1 workgroups per CU (12 / 8 / 4 wavefronts per CU at 256 threads) —
# a test build whose kernels admit up to 1 024 threads (MPRG_KML_THREADS3..5)
out=gpurun_out/r06_c31; mkdir -p $out
export TMPDIR=/tmp
export MPRG_HIP_LIB=$GRAFT_REPO_ROOT/make_prg_amd/_lib/libmprg_hip_thr.so
run() { # name env...
  name=$1; shift
  env "$@" timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_$name.txt 2>&1
  echo "== $name: $(grep '  mprg_kmeans_fit_lds' $out/forest_7500_$name.txt) | $(grep 'device time' $out/forest_7500_$name.txt)"
  env "$@" timeout 600 python tools/forest_profile.py 3750 > $out/forest_3750_$name.txt 2>&1
  echo "   3750: $(grep '  mprg_cluster_loop.small' $out/forest_3750_$name.txt)"
}
run base A=1
run t3_320 MPRG_KML_THREADS3=320
run t4_512 MPRG_KML_THREADS4=512
run t5_1024 MPRG_KML_THREADS5=1024
run t45 MPRG_KML_THREADS4=512 MPRG_KML_THREADS5=1024
run t345 MPRG_KML_THREADS3=320 MPRG_KML_THREADS4=512 MPRG_KML_THREADS5=1024
run t45b MPRG_KML_THREADS4=384 MPRG_KML_THREADS5=512
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
bench() { name=$1; shift
  env "$@" timeout 600 python bench.py $quick > $out/bench_${name}_$RANDOM.json 2> $out/bench_err.txt
  g=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$g').read().strip().splitlines()[-1]); print('30000 $name:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
}
bench base A=1
bench t45 MPRG_KML_THREADS4=512 MPRG_KML_THREADS5=1024
bench t345 MPRG_KML_THREADS3=320 MPRG_KML_THREADS4=512 MPRG_KML_THREADS5=1024
bench base A=1
bench t45 MPRG_KML_THREADS4=512 MPRG_KML_THREADS5=1024
bench t345 MPRG_KML_THREADS3=320 MPRG_KML_THREADS4=512 MPRG_KML_THREADS5=1024
