#!/bin/bash
# round 5, call 55: the wide fits' needed pairs listed per centre, two samples of a centre per lane (KM_WIDE_PAIRS) — beside a test-only build with one
out=gpurun_out/r05_c55; mkdir -p $out
L=$PWD/make_prg_amd/_lib/libmprg_hip_nopairs.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
run() {
  label=$1; size="$2"; shift; shift
  env "$@" MPRG_BACKEND=runtime timeout 900 python tools/deep_profile.py $size --passes 2 > $out/$label.txt 2>&1
  echo "== $label"; grep -E "prg_sha256" $out/$label.txt | tail -1 | cut -c1-150; grep -E "mprg_kmeans_(fit_wide|prepare_big) " $out/$label.txt | head -1
}
run d2k_pairs "2000 4000" X=1
run d2k_single "2000 4000" MPRG_HIP_LIB=$L
run d5k_pairs "5000 10000" X=1
run d5k_single "5000 10000" MPRG_HIP_LIB=$L
run d10k_pairs "10000 20000" X=1
run d10k_single "10000 20000" MPRG_HIP_LIB=$L
