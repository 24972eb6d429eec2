#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02t
mkdir -p $out
run() { tag=$1; shift; "$@" python bench.py --no-cpu-baseline --no-end-to-end $EXTRA > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "batch", d["config"]["batch_per_gpu"])
PY
}
run warm env
run base env
run km128 env MPRG_KM_THREADS=128
run km64 env MPRG_KM_THREADS=64
EXTRA="--batch 15000" run half env
EXTRA="--batch 60000" run double env
EXTRA="--batch 60000 --workers 14" run double14 env
run base2 env
