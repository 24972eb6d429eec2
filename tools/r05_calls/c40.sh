#!/bin/bash
# round 5, call 40: K6's tables by tiles (k_kmeans_prepare_tables_tiled): parity on the GPU, the deep alignments with tiles / with the
# thread-per-element kernel, and with tables for EVERY big problem (MPRG_KM_NO_TABLES_BYTES beyond any problem) instead of on-demand rows
out=gpurun_out/r05_c40; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
run() {
  label=$1; size="$2"; shift; shift
  env "$@" MPRG_BACKEND=runtime timeout 900 python tools/deep_profile.py $size --passes 2 > $out/$label.txt 2>&1
  echo "== $label"; grep -E "prg_sha256" $out/$label.txt | tail -1 | cut -c1-150; grep -E "mprg_kmeans_(fit_wide|prepare_big) " $out/$label.txt | head -2
}
run d2k_tiled "2000 4000" X=1
run d2k_threads "2000 4000" MPRG_KP_TILED=0
run d5k_tiled "5000 10000" X=1
run d5k_alltables "5000 10000" MPRG_KM_NO_TABLES_BYTES=1099511627776
run d10k_tiled "10000 20000" X=1
run d10k_alltables "10000 20000" MPRG_KM_NO_TABLES_BYTES=1099511627776
