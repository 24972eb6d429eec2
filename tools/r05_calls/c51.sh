#!/bin/bash
# round 5, call 51 (experiment, not kept: 7.67 against 7.54 ms per 7 500 alignments): k_cluster_further_one's staging loop with four words in flight
# per lane (a test-only build, -DCF_STAGE_BATCH, of a change that was reverted) beside the product build;
# rocprofv3 kernel trace of one 7 500-alignment engine for the per-launch times of the cluster_further kernels
out=gpurun_out/r05_c51; mkdir -p $out
export TMPDIR=/tmp
L=$PWD/make_prg_amd/_lib/libmprg_hip_cfbatch.so
for v in default batch; do
  lib=$PWD/make_prg_amd/_lib/libmprg_hip.so; [ $v = batch ] && lib=$L
  MPRG_HIP_LIB=$lib MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_$v.txt 2>&1
  echo $v; grep -E "per launch mprg_cluster_further|mprg_cluster_further " $out/profile_$v.txt | cut -c1-200
done
MPRG_BACKEND=runtime rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/forest_profile.py 7500 1 > $out/run_trace.txt 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if r["Kernel_Name"].startswith(("k_cluster_further_one","k_cluster_majority","k_cluster_hamming"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]
for r in rows:
    out.append((r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
n=len(out)//2          # the second of two forests (the profile runs two passes: warm)
for name, us, grid, wg in out[n:n+36]:
    print(f"{name:28s} {us:9.1f} us  grid {grid} wg {wg}")
PY
rm -rf $out/trace
