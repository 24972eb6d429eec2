"""GPU-box helper: what the output directory's filesystem takes (page-cache writes), 1..16 threads, one file or several."""
import os, sys, time, tempfile
from concurrent.futures import ThreadPoolExecutor
import numpy as np
d = sys.argv[1] if len(sys.argv) > 1 else tempfile.gettempdir()
buf = np.random.randint(0, 255, 256 << 20, dtype=np.uint8)
def run(n_files, threads_per_file, gb_per_file=2):
    fds = [os.open(os.path.join(d, f"probe{i}.bin"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC) for i in range(n_files)]
    piece = 4 << 20
    n_piece = (gb_per_file << 30) // piece
    def w(args):
        fd, k0, k1 = args
        mv = memoryview(buf)
        for k in range(k0, k1):
            os.pwrite(fd, mv[(k * piece) % (buf.size - piece):][:piece], k * piece)
    jobs = []
    for fd in fds:
        step = n_piece // threads_per_file
        jobs += [(fd, t * step, (t + 1) * step) for t in range(threads_per_file)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(jobs)) as pool:
        list(pool.map(w, jobs))
    dt = time.perf_counter() - t0
    for fd in fds: os.close(fd)
    for i in range(n_files): os.remove(os.path.join(d, f"probe{i}.bin"))
    print(f"{n_files} file(s) x {threads_per_file} threads: {n_files * gb_per_file / dt:.2f} GB/s", flush=True)
for nf, tp in ((1, 1), (1, 4), (1, 8), (1, 16), (4, 1), (4, 4), (4, 8)):
    run(nf, tp)
