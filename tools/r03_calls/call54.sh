#!/bin/bash
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
python - <<'P'
import json, bench
d=json.load(open('profiles/r03/pmc_summary.json')); print('digest', d['source_digest'], bench.source_digest())
P
