#!/bin/bash
# GPU-box helper: workgroup sizes of the byte kernels on narrow views (exclusive pass, 3 000 alignments)
export TMPDIR=/tmp
out=gpurun_out/r02ag
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
run() { tag=$1; shift; "$@" timeout 600 python bench.py $inproc > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"].replace("mprg_", ""), k["ms"]) for k in r["kernels"][:5]], "verified", d["config"]["verified"]["mismatches"])
PY
}
run cf64 env
run cf128 env MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_cf128.so
run cf256 env MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_cf256.so
run pf128 env MPRG_PF_THREADS=128
run pf64 env MPRG_PF_THREADS=64
run pf512 env MPRG_PF_THREADS=512
run dd128 env MPRG_DD_THREADS=128
run dd64 env MPRG_DD_THREADS=64
run dd512 env MPRG_DD_THREADS=512
