#!/bin/bash
# round 5, call 12: statistics of big problems by many workgroups, more table parts: deep alignments again + parity
out=gpurun_out/r05_c12; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_ddeep.py tests/test_gpu_parity.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for shape in "2000 4000" "5000 10000" "10000 20000"; do
  set -- $shape
  MPRG_DEEP_OUT=$out/deep_$1x$2.json timeout 1200 python tools/deep_profile.py $1 $2 7 --passes 2 > $out/deep_$1x$2.txt 2>&1
  echo "$1 x $2"; grep -E '^\{' $out/deep_$1x$2.txt | tail -1 | cut -c1-220; grep -E "mprg_" $out/deep_$1x$2.txt | head -7
done
