#!/bin/bash
# round 6, call 23: per-launch times of the byte kernels' entry points, level by level (7 500 alignments, one process, one stream)
out=gpurun_out/r06_c23; mkdir -p $out
export TMPDIR=/tmp
MPRG_PROFILE_ALL_LAUNCHES=1 timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500.txt 2>&1
grep "per launch" $out/forest_7500.txt | grep -v kmeans_fit_lds | cut -c1-400
cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/forest_profile.py 7500 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python3 - "$f" > $out/kernel_trace_big.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))) for r in rows]
# the last pass: everything after the last k_ingest... (forest_profile ingests once: take the last third of the trace)
n = len(big); last = big[2 * n // 3:]
for name, us, g, w in last:
    if us >= 150 and "kmeans_fit_lds" not in name:
        print(f"{name:40s} {us:9.1f} us grid {g} wg {w}")
PY
head -80 $out/kernel_trace_big.txt
