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
