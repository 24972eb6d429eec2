#!/bin/bash
# big-problem levels: threshold and width on the deep alignment; the largest problem of config C
out=gpurun_out/r04_c27; mkdir -p $out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/config_c_max_problem.txt
import sys; sys.path.insert(0, '.')
from bench import make_batch
from make_prg_amd.backend import HipBackend
import make_prg_amd.forest as F
F.KM_BIG_BYTES = 0
msas = make_batch(list(range(7500)), 16)[1]
eng = F.ForestEngine(HipBackend(0), 5, 7)
eng.load(msas); eng.run_forest()
print("config C, 7500 alignments: largest clustering problem (count matrix + means, bytes):", eng.counters["max_problem_bytes"])
PY
run() {
  tag=$1; shift
  echo "=== $tag: $*"
  env "$@" timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 2 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-200 | tee $out/deep_$tag.txt
}
run big_256k MPRG_KM_BIG_BYTES=262144
run big_1m MPRG_KM_BIG_BYTES=1048576
run big_4m MPRG_KM_BIG_BYTES=4194304
run big_16m MPRG_KM_BIG_BYTES=16777216
run big_1m_512 MPRG_KM_BIG_BYTES=1048576 MPRG_KM_WIDE_THREADS=512
timeout 900 python -m pytest tests/test_gpu_ddeep.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
