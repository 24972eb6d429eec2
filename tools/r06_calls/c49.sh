#!/bin/bash
# round 6, call 49: the same SQ / memory-side counters for the round's LAST sources (k_kmeans_fit_lds after the chain was shortened)
# (one counter set per pass, --kernel-trace --pmc only; bench.py in-process, one stream, 8 192 alignments per pass)
export TMPDIR=/tmp
out=gpurun_out/r06_c49; mkdir -p $out
inproc="--workers 0 --streams 1 --batch 8192 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 1 --warmup 1"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_WAVES SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$tag -- python3 bench.py $inproc > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  f=$(find $out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $out/counters.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    name = name[5:] if name.startswith("void ") else name
    a = agg[name.split("(")[0]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    if k.startswith(("k_kmeans_restart_select", "k_kmeans_fit_lds", "k_kmeans_prepare_lds", "k_cluster_further_one", "k_partition_wave", "k_ungap_hash")):
        for c, (n, s) in agg[k].items():
            print(k, c, "launches", n, "sum", s, "mean", s / max(n, 1))
PY
  [ -z "$f" ] && echo "$set: no output: $(tail -2 $out/pmc_$tag.err)" | tee -a $out/counters.txt
  rm -rf $out/pmc_$tag
done
