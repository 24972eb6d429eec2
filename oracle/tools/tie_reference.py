"""Container-only (imports the unmodified reference from /root/reference): ties the benchmarked config-C set to the REAL
reference, and says which reference.

(a) `digests`: the real reference (pinned configuration: scikit-learn with n_init=10, OMP_NUM_THREADS=1,
    OPENBLAS_CORETYPE=Haswell) builds config-C seeds 0..n_c-1 and config-B seeds 0..n_b-1; per locus sha256(PRG)[:8] + node
    count.  C records must equal tests/golden/config_c_digests.bin (the oracle's answers, which bench.py and the -m gpu tests
    check the HIP path against); B records must equal the oracle run here.
(b) `stability`: the same reference build of config-C seeds 0..n_s-1 under each of the five OpenBLAS kernel families this NumPy
    ships (exact ties between k-means++ candidates / equidistant centres are decided by the last bits of BLAS
    reductions, whose order differs per family).  A locus is STABLE if its PRG is byte-identical under all five, else
    UNSTABLE; parity is defined against the Haswell-pinned run either way.
Each coretype runs in its own child processes (OPENBLAS_CORETYPE is read when NumPy loads).

Writes tests/golden/config_c_reference_tie.json (read by bench.py: `verified.reference_tie`).

    python -m oracle.tools.tie_reference [--n-c 1000] [--n-b 1000] [--n-s 300] [--procs 8]
    python -m oracle.tools.tie_reference --child CORETYPE CFG START STOP      (internal)"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
FAMILIES = ("Haswell", "SkylakeX", "Sandybridge", "Nehalem", "Prescott")
OUT = os.path.join(ROOT, "tests", "golden", "config_c_reference_tie.json")


def _child(coretype, cfg, start, stop):
    """One process: the real reference on seeds [start, stop) of config cfg; prints hex records."""
    import oracle.refshim.bootstrap as rb
    rb.preset_env(coretype)
    rb.install()
    import tempfile
    from pathlib import Path
    from make_prg.prg_builder import PrgBuilder
    from make_prg_amd.utils.synthetic import synth_config_fasta
    tmp = Path(tempfile.mkdtemp(prefix="mprg_tie_"))
    for seed in range(start, stop):
        p = tmp / f"{cfg}{seed}.fa"
        p.write_text(synth_config_fasta(cfg, seed))
        b = PrgBuilder(f"{cfg}{seed}", p, "fasta", 5, 7)
        prg = b.build_prg()
        rec = hashlib.sha256(prg.encode()).digest()[:8] + struct.pack("<I", b.next_node_id)
        print(seed, rec.hex(), flush=True)
        p.unlink()
    tmp.rmdir()


def reference_records(coretype, cfg, n, procs):
    """{seed: 12-byte record} from `procs` child processes (contiguous seed ranges interleaved for balance)."""
    per = (n + procs * 4 - 1) // (procs * 4)
    ranges = [(s, min(s + per, n)) for s in range(0, n, per)]
    out, running, todo = {}, [], list(ranges)
    while todo or running:
        while todo and len(running) < procs:
            a, b = todo.pop(0)
            running.append(subprocess.Popen([sys.executable, "-m", "oracle.tools.tie_reference", "--child", coretype, cfg, str(a), str(b)],
                                            cwd=ROOT, stdout=subprocess.PIPE, text=True))
        pr = running.pop(0)
        text, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"reference child failed ({coretype} {cfg})")
        for line in text.split("\n"):
            if line:
                seed, hx = line.split()
                out[int(seed)] = bytes.fromhex(hx)
    assert len(out) == n
    return out


def _oracle_b(seed):
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_config_fasta
    prg, b, _ = orc.build_locus_from_text(synth_config_fasta("B", seed), 5, 7)
    return hashlib.sha256(prg.encode()).digest()[:8] + struct.pack("<I", b.next_node_id)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-c", type=int, default=1000)
    ap.add_argument("--n-b", type=int, default=1000)
    ap.add_argument("--n-s", type=int, default=300)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    res = dict(reference="iqbal-lab-org/make_prg v0.5.0 (unmodified, /root/reference) in the build container", made=time.strftime("%Y-%m-%d"),
               pinned=dict(OPENBLAS_CORETYPE="Haswell", OMP_NUM_THREADS=1, n_init=10), record="sha256(PRG)[:8] + uint32 node count")
    import numpy, scipy, sklearn
    res["versions"] = dict(numpy=numpy.__version__, scipy=scipy.__version__, scikit_learn=sklearn.__version__)
    t0 = time.time()
    with open(os.path.join(ROOT, "tests", "golden", "config_c_digests.bin"), "rb") as fh:
        blob = fh.read()
    ref_c = reference_records("Haswell", "C", a.n_c, a.procs)
    bad = [s for s in range(a.n_c) if ref_c[s] != blob[12 * s:12 * s + 12]]
    res["config_c"] = dict(seeds=f"0..{a.n_c - 1}", equal_to_digest_fixture=a.n_c - len(bad), mismatches=bad[:20], seconds=round(time.time() - t0, 1))
    print("config C:", res["config_c"], flush=True)
    assert not bad, "the real reference and tests/golden/config_c_digests.bin disagree"
    if a.n_b:
        t0 = time.time()
        ref_b = reference_records("Haswell", "B", a.n_b, a.procs)
        import multiprocessing as mp
        import oracle.from_msa_oracle as orc
        orc.build_kmeans_lib()
        with mp.get_context("fork").Pool(a.procs) as pool:
            orc_b = pool.map(_oracle_b, range(a.n_b), chunksize=4)
        bad = [s for s in range(a.n_b) if ref_b[s] != orc_b[s]]
        res["config_b"] = dict(seeds=f"0..{a.n_b - 1}", equal_to_oracle=a.n_b - len(bad), mismatches=bad[:20], seconds=round(time.time() - t0, 1))
        print("config B:", res["config_b"], flush=True)
        assert not bad, "the real reference and the oracle disagree on config B"
    if a.n_s:
        t0 = time.time()
        per = {"Haswell": {s: ref_c[s] for s in range(min(a.n_s, a.n_c))}}
        if a.n_s > a.n_c:
            per["Haswell"] = reference_records("Haswell", "C", a.n_s, a.procs)
        for fam in FAMILIES[1:]:
            per[fam] = reference_records(fam, "C", a.n_s, a.procs)
            print(fam, "differs from Haswell on", sum(per[fam][s] != per["Haswell"][s] for s in range(a.n_s)), "of", a.n_s, flush=True)
        unstable = [s for s in range(a.n_s) if len({per[f][s] for f in FAMILIES}) > 1]
        res["stability"] = dict(seeds=f"0..{a.n_s - 1}", families=list(FAMILIES), stable=a.n_s - len(unstable), unstable=len(unstable),
                                differs_from_pinned={f: sum(per[f][s] != per["Haswell"][s] for s in range(a.n_s)) for f in FAMILIES[1:]},
                                unstable_seeds=unstable, seconds=round(time.time() - t0, 1),
                                note="stable = PRG + node count byte-identical under all five OPENBLAS_CORETYPE families; parity of the HIP path "
                                     "and of the oracle is to the Haswell-pinned reference on stable and unstable loci alike")
        print("stability:", {k: v for k, v in res["stability"].items() if k != "unstable_seeds"}, flush=True)
    with open(OUT, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        _child(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    else:
        main()
