#!/bin/bash
# the command line on ONE deep alignment (the ddeep fixture's FASTA): every output type; the PRG against the fixture's hash
out=gpurun_out/r04_c35; mkdir -p $out /tmp/deepcli/in
python - <<'PY'
import json
from make_prg_amd.utils.synthetic import synth_deep_fasta
g = json.load(open('tests/golden/ddeep.json'))
open('/tmp/deepcli/in/ddeep.fa', 'w').write(synth_deep_fasta(g['seed'], g['S'], g['C']))
PY
mkdir -p /tmp/deepcli/out; ( time python -m make_prg_amd from_msa -i /tmp/deepcli/in -o /tmp/deepcli/out/deep -N 7 -L 7 -t 4 -O a --log /tmp/deepcli/log.txt ) 2> $out/time.txt; echo rc=$?
tail -5 $out/time.txt
ls -la /tmp/deepcli/out | tee $out/ls.txt
python - <<'PY' | tee gpurun_out/r04_c35/check.txt
import hashlib, json
g = json.load(open('tests/golden/ddeep.json'))
lines = open('/tmp/deepcli/out/deep.prg.fa').read().split('\n')
prg = lines[1]
print('prg chars', len(prg), 'sha equal to the fixture:', hashlib.sha256(prg.encode()).hexdigest() == g['expect']['prg_sha256'])
PY
tail -5 /tmp/deepcli/log.txt | cut -c1-200
