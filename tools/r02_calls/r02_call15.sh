#!/bin/bash
# GPU-box helper: is the 10-worker step bound by the device or by the host?  worker-count / stream sweep + cgroup throttling
export TMPDIR=/tmp
out=gpurun_out/r02o
mkdir -p $out
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
python bench.py --no-cpu-baseline --no-end-to-end > $out/warm.json 2> $out/warm.err
for cfg in "10 1" "5 1" "14 1" "10 2" "7 2"; do
  set -- $cfg
  echo "workers $1 streams $2"; stat
  python bench.py --no-cpu-baseline --no-end-to-end --workers $1 --streams $2 > $out/w$1_s$2.json 2> $out/w$1_s$2.err
  stat
  python - $out/w$1_s$2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value", d["value"], "ms/step", d["ms_per_step"], "workers", d["config"]["host_worker_processes_per_gpu"], "excl wall/device", d["roofline"]["exclusive_pass"]["wall_ms"], d["roofline"]["exclusive_pass"]["device_ms"])
PY
done
