#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02ap
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
run() { tag=$1; shift; "$@" timeout 600 python bench.py $inproc > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"].replace("mprg_", ""), k["ms"]) for k in r["kernels"][:5]], "verified", d["config"]["verified"]["mismatches"])
PY
}
run default env
run kp256 env MPRG_KP_THREADS=256
run kp512 env MPRG_KP_THREADS=512
run kp1024 env MPRG_KP_THREADS=1024
