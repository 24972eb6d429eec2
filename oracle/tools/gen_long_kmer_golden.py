"""Container-only golden generator: k-mer sizes above 16 (`-L 17`, `-L 20`, `-L 33`; the k-mer size of the clustering IS
min_match_length, recursion_tree.py:453).  Runs the REAL reference (oracle/refshim, pinned configuration of gen_golden.py) on
small hierarchical alignments whose hyper-variable windows are wide enough to be clustered at these lengths, cross-checks the
oracle, and writes tests/golden/long_kmer.json.gz (inputs = generator seeds; expected outputs from the reference).

    python -m oracle.tools.gen_long_kmer_golden"""
import gzip
import json
import sys
import os
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.tools import gen_golden as gg          # installs the reference shim, pins the environment
from make_prg_amd.utils.synthetic import synth_rows_deep


def text_of(seed, S, C, window, period):
    rows = synth_rows_deep(seed, S, C, fanout=(3, 3, 2), rates=(0.6, 0.5, 0.4), window=window, period=period)
    return "".join(f">s{i}\n{r.decode()}\n" for i, r in enumerate(rows))


def main():
    out = []
    for seed, S, C, window, period, N, L in ((1, 60, 600, 90, 200, 5, 17), (2, 80, 800, 120, 260, 5, 20), (3, 70, 900, 150, 300, 4, 33),
                                             (4, 50, 500, 80, 160, 5, 24)):
        text = text_of(seed, S, C, window, period)
        tmp = Path("/tmp/_golden_longk.fa")
        tmp.write_text(text)
        _, ref = gg.run_reference(tmp, N, L)
        mine = gg.run_oracle(text, N, L)
        gg.check_same(f"long k-mer seed {seed} L {L}", ref, mine)
        fits = mine["stats"]["fits"]
        assert fits, "the case must reach KMeans"
        out.append(dict(seed=seed, S=S, C=C, window=window, period=period, N=N, L=L, fasta_sha256=gg.sha(text),
                        expect=gg.pack(ref, full_tree=False), kmeans_fits=len(fits)))
        print("long k-mer", seed, "L", L, "nodes", ref["next_node_id"], "fits", len(fits))
    with gzip.open(os.path.join(ROOT, "tests", "golden", "long_kmer.json.gz"), "wt") as fh:
        json.dump(dict(meta=gg.META, loci=out), fh)


if __name__ == "__main__":
    main()
