#!/bin/bash
# round 5, call 41: as call 40 with the next chunk prefetched in k_kmeans_prepare_tables_tiled and tables for every big problem by default (on-demand rows beside it)
out=gpurun_out/r05_c41; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
run() {
  label=$1; size="$2"; shift; shift
  env "$@" MPRG_BACKEND=runtime timeout 900 python tools/deep_profile.py $size --passes 2 > $out/$label.txt 2>&1
  echo "== $label"; grep -E "prg_sha256" $out/$label.txt | tail -1 | cut -c1-150; grep -E "mprg_kmeans_(fit_wide|prepare_big) " $out/$label.txt | head -2
}
run d2k_tiled "2000 4000" X=1
run d2k_threads "2000 4000" MPRG_KP_TILED=0
run d5k_tiled "5000 10000" X=1
run d5k_ondemand "5000 10000" MPRG_KM_NO_TABLES_BYTES=167772160
run d10k_tiled "10000 20000" X=1
run d10k_ondemand "10000 20000" MPRG_KM_NO_TABLES_BYTES=167772160
