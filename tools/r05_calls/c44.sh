#!/bin/bash
# round 5, call 44: the wide fits' sample-centre distances fetch a sample's whole 64-byte line per trip (KM_XL_LINE) — beside a test-only build without
out=gpurun_out/r05_c44; mkdir -p $out
L=$PWD/make_prg_amd/_lib/libmprg_hip_noline.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
run() {
  label=$1; size="$2"; shift; shift
  env "$@" MPRG_BACKEND=runtime timeout 900 python tools/deep_profile.py $size --passes 2 > $out/$label.txt 2>&1
  echo "== $label"; grep -E "prg_sha256" $out/$label.txt | tail -1 | cut -c1-150; grep -E "mprg_kmeans_(fit_wide|prepare_big) " $out/$label.txt | head -1
}
run d2k_line "2000 4000" X=1
run d2k_noline "2000 4000" MPRG_HIP_LIB=$L
run d5k_line "5000 10000" X=1
run d5k_noline "5000 10000" MPRG_HIP_LIB=$L
run d10k_line "10000 20000" X=1
run d10k_noline "10000 20000" MPRG_HIP_LIB=$L
