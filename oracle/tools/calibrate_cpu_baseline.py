"""Container-only: how fast is the REAL reference compared with the oracle (the CPU "port" that bench.py times on the
GPU box)?  BASELINE.md §3 (2).  Both run here on the same inputs with 8 worker processes, one alignment per task
(the reference's own parallelism, subcommands/from_msa.py:182-183), OMP_NUM_THREADS=1 per worker, PRG construction only
(PrgBuilder(...).build_prg(); no files):
    B     : 1 000 x (50 x 500), seeds 0..999
    C-sub : 500 config-C alignments (~100 x 1-3 kb), seeds 0..499
Writes profiles/r02/cpu_calibration.json; BASELINE.md §3 and bench.py's cpu_baseline.sample quote it.

    python -m oracle.tools.calibrate_cpu_baseline [n_B n_C]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.refshim.bootstrap as rb

rb.preset_env()
rb.install()

import json
import multiprocessing as mp
import tempfile
import time
from pathlib import Path

from make_prg_amd.utils.synthetic import synth_config_fasta

TMP = Path(tempfile.mkdtemp(prefix="mprg_calib_"))


def _path(cfg, seed):
    p = TMP / f"{cfg}_{seed}.fa"
    if not p.exists():
        p.write_text(synth_config_fasta(cfg, seed))
    return p


def _reference_one(args):
    from make_prg.prg_builder import PrgBuilder
    cfg, seed = args
    b = PrgBuilder(f"{cfg}{seed}", _path(cfg, seed), "fasta", 5, 7)
    return len(b.build_prg())


def _port_one(args):
    import oracle.from_msa_oracle as orc
    cfg, seed = args
    return len(orc.build_locus_from_text(_path(cfg, seed).read_text(), 5, 7)[0])


def timed(fn, jobs, procs):
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(fn, jobs[:procs], chunksize=1)          # warm-up: imports, first touch
        t0 = time.perf_counter()
        out = pool.map(fn, jobs, chunksize=1)
        return time.perf_counter() - t0, out


def main():
    n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_c = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    procs = 8
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    res = dict(procs=procs, cpu="build container, 8 vCPU", sets={})
    for name, cfg, n in (("B", "B", n_b), ("C-sub", "C", n_c)):
        jobs = [(cfg, s) for s in range(n)]
        for j in jobs:
            _path(*j)
        t_ref, out_ref = timed(_reference_one, jobs, procs)
        t_port, out_port = timed(_port_one, jobs, procs)
        assert out_ref == out_port, "the port and the reference disagree on PRG lengths"
        res["sets"][name] = dict(n=n, reference_s=round(t_ref, 1), reference_msas_per_s=round(n / t_ref, 2),
                                 port_s=round(t_port, 1), port_msas_per_s=round(n / t_port, 2),
                                 reference_over_port=round(t_port / t_ref, 2))
        print(name, res["sets"][name], flush=True)
    out = Path(ROOT) / "profiles" / "r02" / "cpu_calibration.json"
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps(res, indent=1))
    print("wrote", out)


if __name__ == "__main__":
    main()
