"""The parity-checked DEEP case on the MI355X: one hierarchical alignment of 2 000 x 4 000 (-N 7 -L 7) whose recursion
tree has ~10^4 nodes down to nesting level 6 and ~4 000 KMeans fits of up to 817 sequences x 16 354 k-mers — BASELINE.json
config D's stress (recursion depth, clustering problems far beyond a CU's LDS, the global-memory KMeans kernels) at a size
the oracle finishes in minutes.  Everything is compared with the oracle's fixture (tests/golden/ddeep.json, made by
oracle/tools/gen_ddeep_golden.py): PRG, the product's .bin and .gfa encoders, the full recursion tree, prg_index, counters."""
import json
import os

import numpy as np
import pytest

from tests import parity_common as pc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("fixture,no_tables_from", [("ddeep.json", None), ("ddeep2.json", None), ("ddeep.json", 1), ("ddeep3.json", None)])
def test_deep_hierarchical_alignment_against_the_oracle_fixture(fixture, no_tables_from, monkeypatch):
    """ddeep: 2 000 x 4 000, seed 0, -N 7; ddeep2: 900 x 2 600, seed 3, -N 3 (the nesting limit cuts the recursion short: the
    deepest candidates become multi-allele leaves).  Both fixtures were confirmed by the REAL reference (their "reference" block).
    ddeep3: 700 x 5 200, seed 5, -N 5 — more rows than the LDS row-group table AND more columns than the wide-view rules' 4 096 (32-row
    chunks, three gap-run segments) with clustering below: the big-view kernels of round 4 against the oracle.
    The levels with big clustering problems take mprg_kmeans_fit_wide (forest.KM_BIG_BYTES); the third case also leaves the seeding's
    tables out for them (what levels of 160 MB matrices and more do by default)."""
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.gfa import GFA_Output
    from make_prg_amd.utils.synthetic import synth_deep_fasta
    import make_prg_amd.forest as forest
    if no_tables_from is not None:          # the big levels' problems prepared WITHOUT the seeding's tables: the wide fits compute their rows
        monkeypatch.setattr(forest, "KM_NO_TABLES_BYTES", no_tables_from)
    with open(os.path.join(HERE, "golden", fixture)) as fh:
        g = json.load(fh)
    if "reference" in g:          # (the fixture's own record of the run of the real reference)
        assert g["reference"]["prg_identical"] and g["reference"]["next_node_id_identical"]
    text = synth_deep_fasta(g["seed"], g["S"], g["C"])
    assert pc.sha(text) == g["fasta_sha256"], "the generator changed under the fixture"
    msa = load_alignment_text(text)
    eng = ForestEngine(HipBackend(0), g["N"], g["L"])
    eng.load([msa])
    eng.run_forest()
    prg = eng.assemble_prgs(want_index=True)[0]
    e = g["expect"]
    assert prg is not None and len(prg) == e["prg_len"] and pc.sha(prg) == e["prg_sha256"]
    # the engine skips fits whose result the reference computes and then discards (nesting exhausted, recursion_tree.py:453-456)
    assert (0.9 if g["N"] >= 7 else 0.0) * g["kmeans_fits"] < int(eng.counters["fits"]) <= g["kmeans_fits"]
    assert pc.sha(pc.product_bin_bytes(prg)) == e["bin_sha256"]
    assert pc.sha(GFA_Output.gfa_text(prg)) == e["gfa_sha256"]
    tree = eng.tree_dump(0, msa.ids)
    assert len(tree) == g["nodes"] == e["next_node_id"]
    assert max(n["level"] for n in tree) == max(int(k) for k in g["levels"])
    assert pc.sha(tree) == e["tree_sha256"], "recursion tree differs from the oracle's"
    assert pc.sha(eng.prg_index(0)) == e["prg_index_sha256"]
    assert 5 + 2 * int(eng.site_count[0]) == e["site_num"]


def test_hierarchical_alignment_at_config_d_size_properties():
    """BASELINE.json config D's SIZE with the hierarchical generator (utils/synthetic.synth_rows_deep: 10 000 x 20 000, -N 7 -L 7; the
    recursion nests to the limit: ~1.2 x 10^5 nodes, ~2.7 x 10^4 KMeans fits, a top problem of 10 000 distinct sequences x 16 384
    k-mers whose rounds' fits all go out at once — forest.KM_SPEC_PROBLEMS).  No oracle at this size: what holds for ANY recursion
    tree (tests/prg_walk.py: the site markers nest, every sampled distinct input row is spelt by the PRG — a nested PRG may spell a
    row along more than one path, as the reference's own PRG of `ddeep` does) and the node table's invariants."""
    import time
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import KIND_CLUSTER, KIND_LEAF, ForestEngine
    from make_prg_amd.msa import MSA, Record
    from make_prg_amd.utils.synthetic import synth_rows_deep
    from tests.prg_walk import _prepare, parse_prg, spellings
    rows = synth_rows_deep(0, 10_000, 20_000)
    msa = MSA([Record(r, f"s{i}", f"s{i}") for i, r in enumerate(rows)])
    eng = ForestEngine(HipBackend(0), 7, 7)
    eng.load([msa])
    t0 = time.perf_counter()
    eng.run_forest()
    prg = eng.assemble_prgs()[0]
    wall = time.perf_counter() - t0
    assert prg is not None and eng.counters.get("speculative_levels", 0) >= 1
    # (round 4: 16.6 s; rounds 5 / 6: 2.1 s warm, tools/deep_profile.py — this is the process's FIRST forest: + ~0.5 s of first launches)
    assert wall < 4.0, f"forest + PRG text took {wall:.1f} s: 1.5 x what this build takes"
    tree = parse_prg(prg)                 # (raises unless the markers nest and every site number opens and closes once)
    _prepare(tree)
    distinct = list(dict.fromkeys(r.decode().replace("-", "") for r in rows))
    sample = distinct[::max(1, len(distinct) // 120)]
    paths = [spellings(tree, r) for r in sample]
    assert min(paths) >= 1, f"{sum(p == 0 for p in paths)} of {len(paths)} sampled rows are not spelt by the PRG"
    t, n = eng.tab, eng.n_nodes
    assert n > 50_000 and int(eng.tree_sizes[0]) == n and sorted(eng.node_id.tolist()) == list(range(n))
    kids = np.concatenate([np.arange(f, f + c) for f, c in zip(t["first_child"], t["n_child"]) if c > 0])
    assert sorted(kids.tolist()) == list(range(1, n)), "every node but the root is the child of exactly one node"
    assert (t["n_child"][t["kind"] == KIND_LEAF] == 0).all() and (t["n_child"][t["kind"] != KIND_LEAF] >= 1).all()
    assert (t["n_child"][t["kind"] == KIND_CLUSTER] >= 2).all()
    assert 5 <= int(t["level"].max()) < 7          # nests (nearly) down to the limit


def test_hierarchical_generator_subsamples_against_oracle():
    """Row / column subsamples of that alignment (the same generator stream), -N 7: PRG and node count against the oracle at sizes
    it finishes in seconds — with every level counted as big (threshold 1 byte), so the fits go through the all-rounds-at-once
    launch of wide workgroups and the many-workgroup statistics that the full-size build takes."""
    import oracle.from_msa_oracle as orc
    import make_prg_amd.forest as forest
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.synthetic import synth_rows_deep
    orc.build_kmeans_lib()
    rows = synth_rows_deep(0, 10_000, 20_000)
    texts = []
    for r0, step, nr, c0, nc in ((0, 40, 200, 0, 1500), (3, 25, 300, 9000, 1200), (5000, 1, 150, 4000, 2500)):
        texts.append("".join(f">s{i}\n{rows[i][c0:c0 + nc].decode()}\n" for i in range(r0, r0 + step * nr, step)))
    want = [orc.build_locus_from_text(t, 7, 7) for t in texts]
    for big_bytes in (forest.KM_BIG_BYTES, 1):
        saved = forest.KM_BIG_BYTES
        forest.KM_BIG_BYTES = big_bytes
        try:
            eng = ForestEngine(HipBackend(0), 7, 7)
            eng.load([load_alignment_text(t) for t in texts])
            eng.run_forest()
            prgs = eng.assemble_prgs()
        finally:
            forest.KM_BIG_BYTES = saved
        assert [p for p in prgs] == [w[0] for w in want]
        assert [int(x) for x in eng.tree_sizes] == [w[1].next_node_id for w in want]
        assert big_bytes != 1 or eng.counters.get("speculative_levels", 0) >= 1
