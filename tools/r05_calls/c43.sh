#!/bin/bash
# round 5, call 43: predict() of wide fits by sixteen workgroups per fit (MPRG_KPW_SPLIT): parity, deep alignments with / without
out=gpurun_out/r05_c43; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
run() {
  label=$1; size="$2"; shift; shift
  env "$@" MPRG_BACKEND=runtime timeout 900 python tools/deep_profile.py $size --passes 2 > $out/$label.txt 2>&1
  echo "== $label"; grep -E "prg_sha256" $out/$label.txt | tail -1 | cut -c1-150; grep -E "mprg_kmeans_(fit_wide|prepare_big) " $out/$label.txt | head -2
}
run d2k_split "2000 4000" X=1
run d2k_one "2000 4000" MPRG_KPW_SPLIT=0
run d5k_split "5000 10000" X=1
run d10k_split "10000 20000" X=1
run d10k_one "10000 20000" MPRG_KPW_SPLIT=0
