"""TEST INFRASTRUCTURE: the parity-checked DEEP case "Ddeep" — one hierarchical alignment of 2 000 rows x 4 000 columns
(make_prg_amd.utils.synthetic.synth_rows_deep, seed 0), -N 7 -L 7: ~10^4 recursion-tree nodes down to nesting level 6,
~580 cluster nodes, ~4 000 KMeans fits with up to 817 distinct sequences x 16 354 k-mers (far beyond what fits a CU's LDS)
— BASELINE.json config D's stress (recursion depth, big clustering problems) at a size the oracle finishes in minutes.
Expected values come from the oracle (pinned to the real reference by gen_golden.py); `--reference` additionally runs the
unmodified reference on the same input (container only, slow) and records whether its PRG is identical.
Writes tests/golden/ddeep.json: hashes of the PRG / .bin / .gfa / recursion tree / prg_index + counters.

A second fixture of another seed, shape and nesting limit: `--name ddeep2 --seed 3 --rows 900 --cols 2600 --nesting 3`
(tests/golden/ddeep2.json).

    python -m oracle.tools.gen_ddeep_golden [--reference] [--name NAME --seed S --rows R --cols C --nesting N]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S, C, SEED, N, L = 2000, 4000, 0, 7, 7
NAME = "ddeep"


def _arg(flag, default, conv=int):
    return conv(sys.argv[sys.argv.index(flag) + 1]) if flag in sys.argv else default


def sha(obj):
    if isinstance(obj, str):
        obj = obj.encode()
    elif not isinstance(obj, (bytes, bytearray)):
        obj = json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()
    return hashlib.sha256(obj).hexdigest()


def main():
    global S, C, SEED, N, NAME
    S, C, SEED, N, NAME = _arg("--rows", S), _arg("--cols", C), _arg("--seed", SEED), _arg("--nesting", N), _arg("--name", NAME, str)
    want_reference = "--reference" in sys.argv
    if want_reference:
        import oracle.refshim.bootstrap as rb
        rb.preset_env()
        rb.install()
    import collections
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_deep_fasta
    text = synth_deep_fasta(SEED, S, C)
    fits = []

    def km(M, k):
        fits.append((M.shape[0], M.shape[1], k))
        return orc.kmeans_fit_predict(M, k)

    t0 = time.time()
    prg, b, root = orc.build_locus(orc.load_alignment_text(text), N, L, kmeans=km)
    tree = orc.tree_dump(root)
    out = dict(generator="make_prg_amd.utils.synthetic.synth_deep_fasta", seed=SEED, S=S, C=C, N=N, L=L,
               fasta_sha256=sha(text), oracle_seconds=round(time.time() - t0, 1),
               nodes=len(tree), kinds=dict(collections.Counter(n["kind"] for n in tree)),
               levels={str(k): v for k, v in sorted(collections.Counter(n["level"] for n in tree).items())},
               kmeans_fits=len(fits), max_D=max(f[0] for f in fits), max_V=max(f[1] for f in fits),
               expect=dict(prg_sha256=sha(prg), prg_len=len(prg), bin_sha256=sha(orc.encode_prg_bytes(prg)),
                           gfa_sha256=sha(orc.gfa_text(prg)), tree_sha256=sha(tree), next_node_id=b.next_node_id,
                           site_num=b.site_num,
                           prg_index_sha256=sha(sorted([s, e, n] for (s, e), n in b.prg_index.items()))))
    if want_reference:
        import tempfile
        from pathlib import Path
        from make_prg.prg_builder import PrgBuilder
        p = Path(tempfile.mkdtemp()) / f"{NAME}.fa"
        p.write_text(text)
        t0 = time.time()
        rb_ = PrgBuilder(NAME, p, "fasta", N, L)
        ref_prg = rb_.build_prg()
        out["reference"] = dict(seconds=round(time.time() - t0, 1), prg_identical=ref_prg == prg,
                                next_node_id_identical=rb_.next_node_id == b.next_node_id)
        assert ref_prg == prg, "the real reference disagrees with the oracle on " + NAME
    path = os.path.join(ROOT, "tests", "golden", NAME + ".json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
