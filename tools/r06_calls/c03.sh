#!/bin/bash
# round 6, call 3: wave-instructions per fit of the KMeans kernels (the fits are bound by instruction issue: ~5 cycles of a SIMD per
# wave-instruction explain round 5's kernel times) — SQ instruction counters, forms of round 5 (MPRG_KM_MODE=2) and the LDS form (6)
out=gpurun_out/r06_c03; mkdir -p $out
export TMPDIR=/tmp
inproc="--workers 0 --streams 1 --batch 4096 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 1 --warmup 1"
for mode in ${MODES:-2 6}; do
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  tag=m${mode}_$(echo $set | tr ' ' '_' | cut -c1-40)
  MPRG_KM_MODE=$mode timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$tag -- python3 bench.py $inproc > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  f=$(find $out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $mode <<'PY' | tee -a $out/counters.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    name = name[5:] if name.startswith("void ") else name
    a = agg[name.split("(")[0]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    if k.startswith(("k_kmeans_restart_select", "k_kmeans_fit_lds", "k_cluster_loop")):
        w = agg[k].get("SQ_WAVES", [0, 0])[1]
        for c, (n, s) in agg[k].items():
            print("mode", sys.argv[2], k, c, "launches", n, "sum", s, ("per wave %.1f" % (s / w)) if w else "")
PY
  [ -z "$f" ] && echo "$set: no output: $(tail -2 $out/pmc_$tag.err)" | tee -a $out/counters.txt
  rm -rf $out/pmc_$tag
done
done
