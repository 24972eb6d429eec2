#!/bin/bash
# round 5, call 8: multi-rank command line (gloo, one GPU): build / exchange / place seconds; dedupe wave form re-timed; full default bench
out=gpurun_out/r05_c08; mkdir -p $out
timeout 1200 python tools/multi_rank_timing.py 10000 1,2,4 $out/multi_rank_10000.json > $out/multi_rank.txt 2>&1; cat $out/multi_rank.txt | cut -c1-400
MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_7500.txt 2>&1
grep -E "per launch mprg_(partition|ungap)|device time|mprg_(partition|ungap_dedupe|cluster_further) " $out/profile_7500.txt
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json
