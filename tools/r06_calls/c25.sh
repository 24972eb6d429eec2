#!/bin/bash
# round 6, call 25: the PRG text's copy to pinned host memory is a blit kernel of 256 workgroups that runs beside the next pass's first level
# (kernel trace of c23: its small kernels take 0.2-1.2 ms there instead of 5-50 us) — the runtime's switches for that copy, by `value`
out=gpurun_out/r06_c25; mkdir -p $out
export TMPDIR=/tmp
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $quick > $out/bench_$name.json 2> $out/bench_err_$name.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('30000 $name:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])" || tail -3 $out/bench_err_$name.txt
}
run base A=1
run limit_wg16 DEBUG_CLR_LIMIT_BLIT_WG=16
run limit_wg4 DEBUG_CLR_LIMIT_BLIT_WG=4
run blit_size0 GPU_FORCE_BLIT_COPY_SIZE=0
run sdma1 HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0
run base2 A=1
# what the copy is under each switch: kernel trace of one process (does __amd_rocclr_copyBuffer still show, how long)
for cfg in "base A=1" "limit_wg16 DEBUG_CLR_LIMIT_BLIT_WG=16" "blit_size0 GPU_FORCE_BLIT_COPY_SIZE=0"; do
  set -- $cfg
  ( cd /tmp && rm -rf /tmp/kt_$1 && export $2 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$1 -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/forest_profile.py 7500 > /dev/null 2>&1 )
  f=$(find /tmp/kt_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1"; grep -i "copyBuffer\|k_fr_fill\|k_fr_count\|k_hdr_publish" "$f" | cut -d, -f1-4 | cut -c1-120
done
