#!/bin/bash
# GPU-box helper: k_kmeans_restart at 5 waves per SIMD (96 VGPRs, KM_RMAX 10: 5 workgroups per CU) against the 4-wave build
export TMPDIR=/tmp
out=gpurun_out/r02p
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
timeout 600 python bench.py $inproc > $out/w4.json 2> $out/w4.err
MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_w5.so timeout 600 python bench.py $inproc > $out/w5.json 2> $out/w5.err
python bench.py --no-cpu-baseline --no-end-to-end > $out/b4.json 2> $out/b4.err
MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_w5.so python bench.py --no-cpu-baseline --no-end-to-end > $out/b5.json 2> $out/b5.err
python bench.py --no-cpu-baseline --no-end-to-end > $out/b4b.json 2> $out/b4b.err
MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_w5.so python bench.py --no-cpu-baseline --no-end-to-end > $out/b5b.json 2> $out/b5b.err
python - <<'PY'
import json
for n in ("w4", "w5", "b4", "b5", "b4b", "b5b"):
    try:
        d = json.loads(open(f"gpurun_out/r02p/{n}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, "value", d["value"], "ms/step", d["ms_per_step"], "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:3]], "verified", d["config"]["verified"]["mismatches"])
    except Exception as e:
        print(n, "failed", e)
PY
