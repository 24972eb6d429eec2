"""The reference's small exported helpers (from_msa/cluster_sequences.py:26-208, NodeFactory's private helpers) through
make_prg_amd's functions of the same names, against answers of the REAL reference (tests/golden/helpers.json.gz,
oracle/tools/gen_helpers_golden.py: the reference's own unit-test inputs + seeded random ones).  sequences_are_one_
reference_like / cluster_further run on the device kernel of the recursion (emulation build here, HIP in test_gpu_parity)."""
import copy
import gzip
import json
import os

import numpy as np
import pytest

from make_prg_amd import device
from make_prg_amd.msa import MSA
from tests.emu.backend import EmuBackend

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with gzip.open(os.path.join(HERE, "golden", "helpers.json.gz"), "rt") as fh:
        return json.load(fh)


def outcome(fn, *a, **k):
    try:
        return dict(ok=fn(*a, **k))
    except Exception as e:
        return dict(error=type(e).__name__)


def check_all(g):
    import make_prg_amd.from_msa.cluster_sequences as cs
    from make_prg_amd.recursion_tree import NodeFactory
    want = lambda r: {k: r[k] for k in ("ok", "error") if k in r}
    n = 0
    for r in g["count_distinct_kmers"]:
        assert outcome(cs.count_distinct_kmers, r["seqs"], r["k"]) == want(r); n += 1
    for r in g["count_kmer_occurrences"]:
        assert cs.count_kmer_occurrences(r["seqs"], r["kmers"]).tolist() == r["ok"]; n += 1
    for r in g["get_majority_string"]:
        assert outcome(cs.get_majority_string, r["seqs"]) == want(r); n += 1
    for r in g["hamming_distance"]:
        assert cs.hamming_distance(r["a"], r["b"]) == r["ok"]; n += 1
    for r in g["one_reference_like"]:
        assert cs.sequences_are_one_reference_like(r["seqs"]) == r["ok"], r["seqs"]; n += 1
        assert cs._one_reference_like_host(r["seqs"]) == r["ok"]
    for r in g["cluster_further"]:
        assert cs.cluster_further(r["clusters"]) == r["ok"], r["clusters"]; n += 1
    for r in g["extract_clusters"]:
        assert outcome(cs.extract_clusters, r["seqdict"], r["assign"]) == want(r); n += 1
    for r in g["merge_sequences"]:
        assert outcome(cs.merge_sequences, *r["lists"], first_seq=r["first"]) == want(r); n += 1
    for r in g["merge_clusters"]:
        assert outcome(cs.merge_clusters, *copy.deepcopy(r["clusters"]), first_id=r["first"]) == want(r); n += 1
    for r in g["vertical_partition"]:
        def vp():
            allv, match = NodeFactory._get_vertical_partition(MSA.from_strings(r["rows"]), r["L"])
            return [[[i.start, i.stop] for i in allv], [[i.start, i.stop] for i in match]]
        assert outcome(vp) == want(r), r["rows"]; n += 1
    for r in g["infer_cluster_further"]:
        cr = cs.ClusteringResult([["x"]] * r["n_clusters"])
        assert NodeFactory._infer_if_we_should_cluster_further(MSA.from_strings(r["rows"]), cr, r["level"], r["max_nesting"]) == r["ok"]
        n += 1
    return n


def test_helpers_against_the_reference():
    device.set_backend(EmuBackend())
    try:
        assert check_all(load()) > 300
    finally:
        device.set_backend(None)
