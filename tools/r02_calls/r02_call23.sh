#!/bin/bash
# GPU-box helper: GPU busy % (rocm-smi) and CPUs in use (cgroup usage) while the 10-worker bench runs
export TMPDIR=/tmp
out=gpurun_out/r02v
mkdir -p $out
python bench.py --no-cpu-baseline --no-end-to-end --steps 12 > $out/bench.json 2> $out/bench.err &
pid=$!
sleep 1
: > $out/samples.txt
while kill -0 $pid 2>/dev/null; do
  u=$(rocm-smi --showuse 2>/dev/null | grep -i "GPU use" | head -1 | grep -o "[0-9]*$")
  c=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  echo "$(date +%s.%N) $u $c" >> $out/samples.txt
  sleep 0.2
done
cut -c1-200 $out/bench.json
python - <<'PY'
rows = [l.split() for l in open("gpurun_out/r02v/samples.txt") if len(l.split()) == 3]
t = [float(r[0]) for r in rows]; u = [float(r[1]) for r in rows]; c = [float(r[2]) for r in rows]
n = len(rows)
print("samples", n, "span s", round(t[-1] - t[0], 1))
# last 40 % of the run = the timed steps (generation and ingest come first)
a = int(n * 0.55)
print("GPU use % over the last 45 % of the run: mean", round(sum(u[a:]) / (n - a), 1), "min", min(u[a:]), "max", max(u[a:]))
print("CPUs in use over the same span:", round((c[-1] - c[a]) / 1e6 / (t[-1] - t[a]), 2))
print("GPU use series:", [int(x) for x in u[::max(1, n // 60)]])
PY
