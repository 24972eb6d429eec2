"""GPU-box helper: the command line under 1 / 2 / 4 ranks on the ONE GPU of the box (gloo for the index exchange; RCCL refuses two
ranks on one device), `-O a` on N config-C FASTA files: wall of the run and, per rank, build + write of its segment / index
exchange / placing its bytes in the run's files (utils/segments.py).  The outputs of every run are compared with the one-rank
run's (sha256 of each file).
    python tools/multi_rank_timing.py [files=10000] [ranks=1,2,4] [out.json]"""
import hashlib, json, os, re, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ranks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
out_json = sys.argv[3] if len(sys.argv) > 3 else None
from make_prg_amd.utils.misc import effective_cpus
ncpu = effective_cpus()
root = tempfile.mkdtemp(prefix="mprg_ranks_")
src = os.path.join(root, "msas"); os.mkdir(src)
texts, _ = make_batch(list(range(n_files)), min(16, ncpu))
for sd, t in enumerate(texts):
    with open(os.path.join(src, f"gene{sd:05d}.fa"), "w") as fh:
        fh.write(t)
del texts


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


rows, ref = [], None
for w in ranks:
    outd = os.path.join(root, f"out{w}"); os.mkdir(outd)
    log = os.path.join(root, f"log{w}.txt")
    args = ["from_msa", "-i", src, "-o", os.path.join(outd, "pan"), "-t", str(max(1, ncpu // w)), "-O", "a"]          # (log: stderr)
    env = dict(os.environ, PYTHONPATH=ROOT, MPRG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "make_prg_amd"] + args if w == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(w), "--master-addr", "127.0.0.1", "--master-port",
         str(29600 + w), "-m", "make_prg_amd"] + args
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        print(f"{w} ranks failed:", r.stderr[-1500:]); continue
    text = r.stderr
    per = [tuple(float(x) for x in m) for m in re.findall(r"segments built and written in ([\d.]+)s, index exchange ([\d.]+)s, placed in the run's files in ([\d.]+)s", text)]
    sums = {n: sha(os.path.join(outd, n)) for n in sorted(os.listdir(outd))}
    size = sum(os.path.getsize(os.path.join(outd, n)) for n in sums)
    if ref is None:
        ref = sums
    row = dict(ranks=w, wall_s=round(wall, 2), files_per_s=round(n_files / wall, 1), output_bytes=size, identical_to_first_run=sums == ref,
               build_write_s=[p[0] for p in per], exchange_s=[p[1] for p in per], place_s=[p[2] for p in per], threads_per_rank=max(1, ncpu // w))
    rows.append(row)
    print(json.dumps(row), flush=True)
    shutil.rmtree(outd, ignore_errors=True)
shutil.rmtree(root, ignore_errors=True)
if out_json:
    json.dump(dict(files=n_files, cpus=ncpu, runs=rows), open(out_json, "w"), indent=1)
