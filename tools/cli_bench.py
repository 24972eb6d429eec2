"""GPU-box helper: end-to-end wall time of the command line (files on disk -> .prg.fa / .bin.zip / .gfa.zip / update_DS.zip)
for N synthetic config-C alignments.  usage: cli_bench.py [N] [threads] [output types ...]; an output type may carry environment
settings for its run: a:MPRG_WRITE_THREADS=1,MPRG_CHUNK=2048"""
import os
import subprocess
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprocessing import Pool
from make_prg_amd.utils.synthetic import synth_config_fasta


def _write(args):
    d, s = args
    with open(os.path.join(d, f"gene{s}.fa"), "w") as fh:
        fh.write(synth_config_fasta("C", s))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    t = sys.argv[2] if len(sys.argv) > 2 else "10"
    types = sys.argv[3:] or ["p", "a"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "msas")
        os.mkdir(d)
        with Pool(32) as pool:
            pool.map(_write, [(d, s) for s in range(n)], chunksize=16)
        for run, spec in enumerate(types):
            ot, _, envs = spec.partition(":")
            env = dict(os.environ, **dict(kv.split("=", 1) for kv in envs.split(",") if kv))
            t0 = time.time()
            res = subprocess.run([sys.executable, "-m", "make_prg_amd", "from_msa", "-i", d, "-o", os.path.join(tmp, f"out_{run}", "pan"),
                                  "-t", t, "-O", ot, "--log", os.path.join(tmp, "log.txt")], cwd=root, capture_output=True, text=True, env=env)
            dt = time.time() - t0
            import shutil
            shutil.rmtree(os.path.join(tmp, f"out_{run}"), ignore_errors=True)
            print(f"-O {ot} {envs} -t {t}: {n} files in {dt:.1f}s = {n / dt:.0f} loci/s (rc {res.returncode}) {res.stderr[-300:] if res.returncode else ''}", flush=True)
            print("".join(l for l in res.stderr.splitlines(True) if l.startswith("[pipeline]")), end="")
            print("   " + " | ".join(l.split(":", 2)[-1].strip() for l in open(os.path.join(tmp, "log.txt")) if "built in" in l or "written in" in l)[-300:])
