"""BASELINE.json config B at full size on the GPU: 1 000 synthetic alignments (50 x 500, seeds 0..999, -N 5 -L 7) in
ONE resident batch, every PRG compared with the oracle (run in worker processes before the GPU is touched), plus a
config-C sample.  Size-independent checks on top: every PRG round-trips through the binary encoder and its site
markers pair up."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(args):
    cfg, seed = args
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_config_fasta
    prg, b, root = orc.build_locus_from_text(synth_config_fasta(cfg, seed), 5, 7)
    return prg, b.next_node_id


def _expected(cfg, seeds):
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    with mp.get_context("fork").Pool(min(os.cpu_count() or 1, 64)) as pool:
        return pool.map(_oracle, [(cfg, s) for s in seeds], chunksize=4)


CASES = [("B", 0, 1000), ("C", 5000, 5128)]


@pytest.fixture(scope="module")
def oracle_results():
    """All expectations first: the worker processes are forked before this process makes its first HIP call
    (this file sorts before the other GPU test files)."""
    return {(cfg, lo, hi): _expected(cfg, list(range(lo, hi))) for cfg, lo, hi in CASES}


def run_gpu(cfg, seeds):
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.synthetic import synth_config_fasta
    eng = ForestEngine(HipBackend(0), 5, 7)
    eng.load([load_alignment_text(synth_config_fasta(cfg, s)) for s in seeds])
    eng.run_forest()
    prgs = eng.assemble_prgs()
    n_nodes = np.bincount(eng.tab["msa"], minlength=len(seeds))
    return prgs, n_nodes


def check_markers(prg):
    units = prg.split()
    markers = [int(u) for u in units if u.isdigit()]
    odd = [m for m in markers if m % 2 == 1]
    assert all(odd.count(m) == 2 for m in set(odd)), "every site opens and closes once"
    assert all(u for u in units)


@pytest.mark.parametrize("cfg,lo,hi", CASES)
def test_full_config_against_oracle(oracle_results, cfg, lo, hi):
    seeds = list(range(lo, hi))
    want = oracle_results[(cfg, lo, hi)]
    prgs, n_nodes = run_gpu(cfg, seeds)
    bad = [s for s, p, (w, _) in zip(seeds, prgs, want) if p != w]
    assert not bad, f"{len(bad)} of {len(seeds)} loci differ from the oracle, first seeds {bad[:5]}"
    assert [int(x) for x in n_nodes] == [n for _, n in want]
    for p in prgs[::25]:
        check_markers(p)


def test_config_d_at_full_size_properties(oracle_results):
    """BASELINE.json config D: ONE alignment of 10 000 x 20 000 (200 MB per copy), -N 7 -L 7.  No oracle at this size:
    checks that hold for ANY recursion tree (tests/prg_walk.py) — the site markers nest, every site number is used once, and
    every distinct ungapped input row is spelt by exactly ONE choice of alleles through the PRG — plus the node table's own
    invariants (children contiguous, every node reachable once, leaves partition the root's columns along every path)."""
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import KIND_LEAF, ForestEngine
    from make_prg_amd.msa import MSA, Record
    from make_prg_amd.utils.synthetic import synth_rows
    from tests.prg_walk import check_prg_spells_rows
    rows = synth_rows(0, 10_000, 20_000, 8)
    msa = MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(rows)])
    eng = ForestEngine(HipBackend(0), 7, 7)
    eng.load([msa])
    eng.run_forest()
    prg = eng.assemble_prgs()[0]
    assert prg is not None and len(prg) > 10_000 * 19_000
    n_rows = check_prg_spells_rows(prg, [r.decode() for r in rows])
    assert n_rows == len({r.replace(b"-", b"") for r in rows})
    t = eng.tab
    n = eng.n_nodes
    assert n >= 2 and int(eng.tree_sizes[0]) == n and sorted(eng.node_id.tolist()) == list(range(n))
    kids = np.concatenate([np.arange(f, f + c) for f, c in zip(t["first_child"], t["n_child"]) if c > 0])
    assert sorted(kids.tolist()) == list(range(1, n)), "every node but the root is the child of exactly one node"
    from make_prg_amd.forest import KIND_CLUSTER
    assert (t["n_child"][t["kind"] == KIND_LEAF] == 0).all() and (t["n_child"][t["kind"] != KIND_LEAF] >= 1).all()
    assert (t["n_child"][t["kind"] == KIND_CLUSTER] >= 2).all()          # (a tree root may be a MultiIntervalNode with one child)
    assert int(t["level"].max()) < 7


def _oracle_text(args):
    text, N = args
    import oracle.from_msa_oracle as orc
    prg, b, root = orc.build_locus_from_text(text, N, 7)
    return prg, b.next_node_id


def test_config_d_generator_subsamples_against_oracle():
    """Row / column subsamples of the config-D alignment (the same generator stream), -N 7: PRG and node count against the
    oracle, at sizes it finishes in seconds."""
    import oracle.from_msa_oracle as orc
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.synthetic import synth_rows
    orc.build_kmeans_lib()
    rows = synth_rows(0, 10_000, 20_000, 8)
    texts = []
    for r0, nr, c0, nc in ((0, 300, 0, 3000), (5000, 200, 12000, 5000), (17, 400, 400, 1500)):
        texts.append("".join(f">s{i}\n{rows[i][c0:c0 + nc].decode()}\n" for i in range(r0, r0 + nr)))
    want = [_oracle_text((t, 7)) for t in texts]
    eng = ForestEngine(HipBackend(0), 7, 7)
    eng.load([load_alignment_text(t) for t in texts])
    eng.run_forest()
    prgs = eng.assemble_prgs()
    assert [p for p in prgs] == [w[0] for w in want]
    assert [int(x) for x in eng.tree_sizes] == [w[1] for w in want]
