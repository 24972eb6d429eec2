"""Re-entry of NodeFactory.build below an existing node (what the reference's `update` does per touched leaf,
recursion_tree.py:352-391): sub-trees, node ids and nesting levels against vectors produced by the real reference
(oracle/tools/gen_reentry_golden.py), one by one and as ONE batch (NodeFactory.build_many)."""
import gzip
import json
import os

import pytest

from make_prg_amd import device
from make_prg_amd.msa import MSA
from make_prg_amd.prg_builder import PrgBuilder
from make_prg_amd.recursion_tree import LeafNode, MultiClusterNode, NodeFactory
from tests.emu.backend import EmuBackend

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cases():
    with gzip.open(os.path.join(HERE, "golden", "reentry.json.gz"), "rt") as fh:
        return json.load(fh)["cases"]


@pytest.fixture(autouse=True, scope="module")
def _emu():
    device.set_backend(EmuBackend())
    yield
    device.set_backend(None)


class _Parent:
    """Stand-in for the existing parent node: the re-entry only reads its nesting level and id."""

    def __init__(self, level, node_id):
        self.nesting_level, self.node_id = level, node_id


def _builder(case, first_id):
    b = PrgBuilder(case["file"], None, "fasta", case["N"], case["L"], _root_factory=lambda self: None)
    b.next_node_id = first_id
    return b


def _dump(root):
    out = []

    def rec(n):
        kind = "leaf" if isinstance(n, LeafNode) else ("cluster" if isinstance(n, MultiClusterNode) else "interval")
        out.append(dict(id=n.node_id, kind=kind, level=n.nesting_level, parent=None if n.parent is None else n.parent.node_id,
                        rows=[[r.id, str(r.seq)] for r in n.alignment], children=[c.node_id for c in n.children]))
        for c in n.children:
            rec(c)
    rec(root)
    return out


def _jobs(case, builder):
    return [(MSA.from_strings([s for _, s in j["rows"]], [i for i, _ in j["rows"]], [i for i, _ in j["rows"]]), builder,
             None if j["parent_level"] is None else _Parent(j["parent_level"], j["parent_id"])) for j in case["jobs"]]


def test_one_by_one(cases):
    n_nontrivial = 0
    for case in cases:
        b = _builder(case, case["first_node_id"])
        for (aln, builder, parent), j in zip(_jobs(case, b), case["jobs"]):
            sub = NodeFactory.build(aln, builder, parent)
            assert _dump(sub) == j["subtree"], case["file"]
            assert b.next_node_id == j["next_node_id"]
            n_nontrivial += len(j["subtree"]) > 1
    assert n_nontrivial >= 20


def test_all_touched_leaves_of_all_loci_in_one_batch(cases):
    builders = [_builder(c, c["first_node_id"]) for c in cases]
    jobs, owner = [], []
    for c, b in zip(cases, builders):
        for job in _jobs(c, b):
            jobs.append(job)
            owner.append(c)
    subs = NodeFactory.build_many(jobs)
    k = 0
    for c, b in zip(cases, builders):
        for j in c["jobs"]:
            assert _dump(subs[k]) == j["subtree"], c["file"]
            k += 1
        assert b.next_node_id == c["jobs"][-1]["next_node_id"]
