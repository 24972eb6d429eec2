#!/bin/bash
# round 5, call 27: config D (flat 10 000 x 20 000) after batched loads in k_ingest / k_ungap_hash / k_gap_runs / k_cluster_majority,
# candidate classes in k_dedupe_scan_big, group-per-row k_cluster_hamming; k_partition's phases on that alignment
out=gpurun_out/r05_c27; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_b.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 600 python tools/config_d_profile.py --passes 4 > $out/config_d_timing.txt 2>&1; tail -22 $out/config_d_timing.txt | cut -c1-200
timeout 600 python tools/phase_timing.py flat 10000 20000 > $out/phase_flat.txt 2>&1; tail -12 $out/phase_flat.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv; head -16 $f | cut -c1-160
rm -rf $out/prof
