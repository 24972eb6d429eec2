"""GPU-box helper: host memory of one bench worker process (RSS after a step) and the box's CPUs / memory."""
import os
import subprocess
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil

print("cpus", os.cpu_count(), "mem GiB", round(psutil.virtual_memory().total / 2**30), "available GiB",
      round(psutil.virtual_memory().available / 2**30))
p = subprocess.Popen([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "6"], stdout=subprocess.PIPE, text=True)
peak = {}
while p.poll() is None:
    try:
        for c in psutil.Process(p.pid).children(recursive=True):
            peak[c.pid] = max(peak.get(c.pid, 0), c.memory_info().rss)
    except psutil.Error:
        pass
    time.sleep(0.5)
print("worker peak RSS GiB:", sorted(round(v / 2**30, 2) for v in peak.values()))
print(p.stdout.read()[:200])
