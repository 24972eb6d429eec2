"""GPU-box helper: filling ONE new file through a shared mapping with 1..16 threads (page faults per page, no inode lock
held across the copy) against pwrite from the same threads — is a mapped .bin.zip faster than the 10 GB/s of buffered writes?"""
import mmap, os, sys, time, tempfile
from concurrent.futures import ThreadPoolExecutor
import numpy as np
d = sys.argv[1] if len(sys.argv) > 1 else tempfile.gettempdir()
src = np.random.randint(0, 255, 256 << 20, dtype=np.uint8)
GB = 2
def run_mmap(threads, prealloc):
    path = os.path.join(d, "probe_mmap.bin")
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    size = GB << 30
    t0 = time.perf_counter()
    if prealloc:
        os.posix_fallocate(fd, 0, size)
    else:
        os.ftruncate(fd, size)
    mm = mmap.mmap(fd, size)
    dst = np.frombuffer(mm, np.uint8)
    piece = 4 << 20
    n = size // piece
    def w(t):
        for k in range(t, n, threads):
            dst[k * piece:(k + 1) * piece] = src[(k * piece) % (src.size - piece):][:piece]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(w, range(threads)))
    dt = time.perf_counter() - t0
    del dst
    mm.close(); os.close(fd); os.remove(path)
    print(f"mmap {'fallocate' if prealloc else 'ftruncate'} x {threads} threads: {GB / dt:.2f} GB/s", flush=True)
for th in (1, 4, 8, 16):
    run_mmap(th, False)
for th in (4, 16):
    run_mmap(th, True)
