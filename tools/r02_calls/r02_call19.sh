#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02s
mkdir -p $out
run() { tag=$1; shift; "$@" python bench.py --no-cpu-baseline --no-end-to-end $EXTRA > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"])
PY
}
run warm env
for rep in 1 2; do
run base_$rep env
run q1_$rep env GPU_MAX_HW_QUEUES=1
run q2_$rep env GPU_MAX_HW_QUEUES=2
run nosdma_$rep env HSA_ENABLE_SDMA=0
EXTRA="--workers 7 --streams 2" run w7s2_$rep env
EXTRA="--workers 8 --streams 2" run w8s2_$rep env GPU_MAX_HW_QUEUES=2
EXTRA=""
done
