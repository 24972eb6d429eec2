#!/bin/bash
# round 5, call 17: update_DS members with the alignment at four bits per cell: CLI parity tests + command-line rates (-O p, -O a), packed and not
out=gpurun_out/r05_c17; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_update.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 1200 python tools/cli_bench.py 30000 16 p a p:MPRG_PACK_ALIGNMENTS=0 a:MPRG_PACK_ALIGNMENTS=0 p a > $out/cli.txt 2>&1
grep -E "^-O" $out/cli.txt
