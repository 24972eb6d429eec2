#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02z
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
for v in base cb16_eb8 cb16_eb16 cb4_eb8; do
  if [ $v = base ]; then unset MPRG_HIP_LIB; else export MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_$v.so; fi
  timeout 600 python bench.py $inproc > $out/$v.json 2> $out/$v.err
  python - $out/$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:2]], "verified", d["config"]["verified"]["mismatches"])
PY
done
