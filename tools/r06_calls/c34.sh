#!/bin/bash
# round 6, call 34: how busy the device is while `value` is measured (four worker processes): rocm-smi's busy percentage sampled beside bench.py
out=gpurun_out/r06_c34; mkdir -p $out
export TMPDIR=/tmp
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 120"
python bench.py $quick > $out/bench.json 2> $out/bench_err.txt &
bp=$!
sleep 12          # (generation + ingest + warm-up)
for i in $(seq 1 80); do
  kill -0 $bp 2>/dev/null || break
  rocm-smi --showuse --showmemuse --json 2>/dev/null | tr -d '\n' >> $out/smi.txt; echo >> $out/smi.txt
  sleep 0.25
done
wait $bp
python - <<'PY'
import json
vals=[]
for line in open('gpurun_out/r06_c34/smi.txt'):
    line=line.strip()
    if not line.startswith('{'): continue
    try:
        d=json.loads(line)
        for k,v in d.items():
            if isinstance(v,dict):
                for kk,vv in v.items():
                    if 'GPU use' in kk: vals.append(float(vv))
    except Exception as e: pass
print('samples', len(vals), 'busy % mean', sum(vals)/max(len(vals),1), 'min', min(vals or [0]), 'max', max(vals or [0]))
d=json.loads(open('gpurun_out/r06_c34/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
PY
head -c 600 $out/smi.txt
