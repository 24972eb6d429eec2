"""Container-only: known answers of the reference's small exported helpers (make_prg/from_msa/cluster_sequences.py:26-208
and NodeFactory's private helpers, recursion_tree.py:475-572), produced by calling the REAL reference functions on the
inputs of its own unit tests (tests/from_msa/test_cluster_sequences.py:94-300 and kin) plus seeded random inputs.
Writes tests/golden/helpers.json.gz; tests/test_helpers_api.py replays it through make_prg_amd's functions.

    python -m oracle.tools.gen_helpers_golden"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.refshim.bootstrap as rb

rb.preset_env()
rb.install()

import gzip
import json
import random

from Bio.AlignIO import MultipleSeqAlignment
from Bio.Seq import Seq
from Bio.SeqRecord import SeqRecord
import make_prg.from_msa.cluster_sequences as cs
from make_prg.recursion_tree import NodeFactory

OUT = os.path.join(ROOT, "tests", "golden", "helpers.json.gz")


def call(fn, *a, **k):
    try:
        return dict(ok=fn(*a, **k))
    except Exception as e:                       # the error TYPE is part of the contract
        return dict(error=type(e).__name__)


def msa(rows, ids=None):
    ids = ids or [f"s{i}" for i in range(len(rows))]
    return MultipleSeqAlignment([SeqRecord(Seq(r), id=i, description=i) for r, i in zip(rows, ids)])


def main():
    rng = random.Random(11)
    rec = dict(count_distinct_kmers=[], count_kmer_occurrences=[], get_majority_string=[], hamming_distance=[],
               one_reference_like=[], cluster_further=[], extract_clusters=[], merge_sequences=[], merge_clusters=[],
               vertical_partition=[], infer_cluster_further=[])
    # ---- the reference's own unit-test inputs
    for seqs, k in ((["AAA"], 5), (["AAAAAAAAAT"], 5), (["AAAT", "AAAG"], 3)):
        rec["count_distinct_kmers"].append(dict(seqs=seqs, k=k, **call(cs.count_distinct_kmers, seqs, k)))
    for seqs, kmers in ((["AAAAT"], {"AAA": 0, "AAT": 1}), (["AAAAT", "AAATA"], {"AAA": 0, "AAT": 1, "ATA": 2})):
        rec["count_kmer_occurrences"].append(dict(seqs=seqs, kmers=kmers, ok=cs.count_kmer_occurrences(seqs, kmers).tolist()))
    unit_sets = [["AATA", "AAAA", "AAGA", "AATA"], ["ATTT", "TTTT"], ["AAAAA", "AAAAT", "TTTTT"], ["AA", "AT", "TT", "CC"],
                 ["A-AT", "AAAT", "A-AT"], ["AAAAAAAAAA", "AAAAAAAATT", "AAAAAAATTT"], ["ACGT"], ["AC", "ACG"]]
    for seqs in unit_sets:
        rec["get_majority_string"].append(dict(seqs=seqs, **call(cs.get_majority_string, seqs)))
        if len({len(s) for s in seqs}) == 1:
            rec["one_reference_like"].append(dict(seqs=seqs, ok=cs.sequences_are_one_reference_like(seqs)))
    for a, b in (("AATTA", "AATTA"), ("AATTA", "AATAA"), ("AAAAA", "TTTTT"), ("A-A", "AAA")):
        rec["hamming_distance"].append(dict(a=a, b=b, ok=cs.hamming_distance(a, b)))
    for seqdict, assign in (({"AAA": ["s1", "s2"], "AAT": ["s3"], "TTT": ["s4"]}, [0, 0, 1]),
                            ({"AAA": ["s1"], "AAT": ["s3"]}, [0, 1, 1]), ({"AAA": ["s1"], "AAT": ["s3"]}, [0, 2]),
                            ({"AAA": ["AAA", "A-A"], "TTT": ["TTT"]}, [1, 0])):
        rec["extract_clusters"].append(dict(seqdict=seqdict, assign=assign, **call(cs.extract_clusters, seqdict, assign)))
    for lists, first in (([["AAA", "AAT"], ["TTT"]], "AAT"), ([["ARA", "AAT"]], "ARA"), ([["AAA"], ["AAA", "CCC"]], "AAA"),
                         ([["AAA"]], "TTT")):
        rec["merge_sequences"].append(dict(lists=lists, first=first, **call(cs.merge_sequences, *lists, first_seq=first)))
    for clusters, first in (([[["s1", "s2"], ["s3"]], [["s4"]]], "s3"), ([[["s1", "s2"], ["s3"]]], "s2"), ([[["s1"]]], "s9")):
        import copy
        rec["merge_clusters"].append(dict(clusters=clusters, first=first, **call(cs.merge_clusters, *copy.deepcopy(clusters), first_id=first)))
    # ---- seeded random inputs
    for _ in range(60):
        n, w = rng.randint(1, 9), rng.randint(1, 30)
        base = [rng.choice("ACGT") for _ in range(w)]
        seqs = []
        for _ in range(n):
            s = list(base)
            for _ in range(rng.randint(0, max(1, w // 3))):
                s[rng.randrange(w)] = rng.choice("ACGT-")
            seqs.append("".join(s))
        rec["get_majority_string"].append(dict(seqs=seqs, ok=cs.get_majority_string(seqs)))
        rec["one_reference_like"].append(dict(seqs=seqs, ok=cs.sequences_are_one_reference_like(seqs)))
    for _ in range(40):
        clusters = []
        w = rng.randint(2, 25)
        for _ in range(rng.randint(1, 4)):
            base = [rng.choice("ACGT") for _ in range(w)]
            cl = []
            for _ in range(rng.randint(1, 6)):
                s = list(base)
                for _ in range(rng.randint(0, max(1, w // 3))):
                    s[rng.randrange(w)] = rng.choice("ACGT")
                cl.append("".join(s))
            clusters.append(cl)
        rec["cluster_further"].append(dict(clusters=clusters, ok=cs.cluster_further(clusters)))
    for _ in range(40):
        S, C = rng.randint(1, 12), rng.randint(1, 60)
        base = [rng.choice("ACGT") for _ in range(C)]
        rows = []
        for _ in range(S):
            s = list(base)
            for _ in range(rng.randint(0, 4)):
                p = rng.randrange(C)
                s[p] = rng.choice("ACGT-")
            rows.append("".join(s))
        L = rng.choice((1, 3, 7))
        try:
            allv, match = NodeFactory._get_vertical_partition(msa(rows), L)
            got = dict(ok=[[[i.start, i.stop] for i in allv], [[i.start, i.stop] for i in match]])
        except Exception as e:
            got = dict(error=type(e).__name__)
        rec["vertical_partition"].append(dict(rows=rows, L=L, **got))
        for n_clusters, level, max_nesting in ((1, 0, 5), (2, 0, 5), (2, 4, 5), (3, 1, 2)):
            cr = cs.ClusteringResult([["x"]] * n_clusters)
            rec["infer_cluster_further"].append(dict(rows=rows, n_clusters=n_clusters, level=level, max_nesting=max_nesting,
                                                     ok=NodeFactory._infer_if_we_should_cluster_further(msa(rows), cr, level, max_nesting)))
    with gzip.open(OUT, "wt") as fh:
        json.dump(dict(meta=dict(reference="iqbal-lab-org/make_prg v0.5.0, real functions under oracle/refshim"), **rec), fh,
                  separators=(",", ":"))
    print("wrote", OUT, {k: len(v) for k, v in rec.items()})


if __name__ == "__main__":
    main()
