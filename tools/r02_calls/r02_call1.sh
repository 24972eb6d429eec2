#!/bin/bash
# round-2 GPU call 1: (1) config-C digest fixture by the oracle on the box's host cores, (2) the HIP path against it at
# full size, (3) k_column_masks on config D's root view under rocprofv3 (kernel stats, then FETCH_SIZE in its own pass)
export TMPDIR=/tmp
out=gpurun_out/r02a
mkdir -p $out
python oracle/tools/gen_config_c_digests.py --out $out/config_c_digests.bin > $out/digests.log 2>&1
cp $out/config_c_digests.bin tests/golden/config_c_digests.bin
python tests/config_c_full.py 0 30000 7500 > $out/config_c_full.json 2> $out/config_c_full.err
echo "config_c_full rc=$?" >> $out/digests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/masks_prof -- python3 tools/masks_bench.py > $out/masks_bench.txt 2> $out/masks_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/masks_pmc_fetch -- python3 tools/masks_bench.py > $out/masks_bench_pmc.txt 2> $out/masks_pmc.err
cat $out/digests.log $out/config_c_full.json $out/masks_bench.txt
find $out -name "*.csv" | head
