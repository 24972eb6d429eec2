"""Container-only: golden vectors for the RE-ENTRY of NodeFactory.build below an existing node (what the reference's
`update` does for every touched leaf, recursion_tree.py:352-391), produced by the real reference.

For a few loci of the reference's integration inputs: build the locus, take its leaves with the most sequences, add one
new row to each leaf's alignment (the leaf's first row with a few substitutions and one deletion — a stand-in for the
aligner's output, which is an input here), call the reference's NodeFactory.build(updated, builder, leaf.parent) in
order, and record inputs + the sub-tree each call returns (+ the builder's next_node_id after each).

    python -m oracle.tools.gen_reentry_golden
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.refshim.bootstrap as rb
rb.preset_env()
rb.install()

import gzip, json
from pathlib import Path

from Bio.AlignIO import MultipleSeqAlignment
from Bio.SeqRecord import SeqRecord
from Bio.Seq import Seq
from make_prg.prg_builder import PrgBuilder
from make_prg.recursion_tree import NodeFactory, LeafNode, MultiClusterNode, MultiIntervalNode

DATA = Path("/root/reference/tests/integration_tests/data")
OUT = Path(ROOT) / "tests" / "golden" / "reentry.json.gz"
FILES = ["sample_example/GC00006032.fa", "sample_example/GC00010897.fa", "nested_snps_seq_backgrounds_more_seqs.fa",
         "amira_MSAs/" + sorted(os.listdir(DATA / "amira_MSAs"))[0] if (DATA / "amira_MSAs").is_dir() else "match.fa",
         "synthetic:B:3", "synthetic:B:11"]


def tree_dump(root):
    out = []

    def rec(n):
        kind = "leaf" if isinstance(n, LeafNode) else ("cluster" if isinstance(n, MultiClusterNode) else "interval")
        out.append(dict(id=n.node_id, kind=kind, level=n.nesting_level, parent=None if n.parent is None else n.parent.node_id,
                        rows=[[r.id, str(r.seq)] for r in n.alignment], children=[c.node_id for c in n.children]))
        for c in n.children:
            rec(c)
    rec(root)
    return out


def leaves(n):
    return [n] if isinstance(n, LeafNode) else [l for c in n.children for l in leaves(c)]


def new_row(seq: str, salt: int) -> str:
    """The leaf's first row with a block of substitutions in the middle, scattered substitutions and one deletion."""
    s = list(seq)
    pos = [i for i, c in enumerate(s) if c in "ACGT"]
    for j, p in enumerate(pos[salt % 3::max(len(pos) // 5, 1)][:5]):
        s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + j) % 4]
    mid = len(pos) // 2
    for j, p in enumerate(pos[mid:mid + 6]):
        s[p] = "ACGT"[("ACGT".index(s[p]) + 2) % 4]
    if len(pos) > 12:
        s[pos[len(pos) // 4]] = "-"
    return "".join(s)


def main():
    cases = []
    import tempfile
    from make_prg_amd.utils.synthetic import synth_config_fasta
    tmp = tempfile.mkdtemp()
    for name in FILES:
        if name.startswith("synthetic:"):
            _, cfg, seed = name.split(":")
            path = Path(tmp) / f"synth_{cfg}_{seed}.fa"
            path.write_text(synth_config_fasta(cfg, int(seed)))
        else:
            path = DATA / name
        if not path.exists():
            print("skip (absent)", name)
            continue
        for N, L in ((5, 7), (3, 3)):
            b = PrgBuilder(path.stem, path, "fasta", N, L)
            b.build_prg()
            # the widest leaves with several sequences: their re-entry gives real sub-trees
            picked = sorted(leaves(b.root), key=lambda l: (-len(l.alignment) * l.alignment.get_alignment_length(), l.node_id))[:4]
            jobs = []
            for salt, leaf in enumerate(picked):
                rows = [[r.id, str(r.seq)] for r in leaf.alignment]
                rows.append([f"denovo_{salt}", new_row(rows[0][1], salt)])
                if salt % 2 and len(rows) > 2:
                    rows.append([f"denovo_{salt}b", new_row(rows[1][1], salt + 1)])
                updated = MultipleSeqAlignment([SeqRecord(Seq(s), id=i, description=i) for i, s in rows])
                parent = leaf.parent
                sub = NodeFactory.build(updated, b, parent)
                jobs.append(dict(rows=rows, parent_level=None if parent is None else parent.nesting_level,
                                 parent_id=None if parent is None else parent.node_id,
                                 subtree=tree_dump(sub), next_node_id=b.next_node_id))
            cases.append(dict(file=name, N=N, L=L, first_node_id=jobs[0]["subtree"][0]["id"] if jobs else None, jobs=jobs))
            print(name, N, L, [(j["subtree"][0]["kind"], len(j["subtree"])) for j in jobs])
    with gzip.open(OUT, "wt") as fh:
        json.dump(dict(meta=dict(reference="iqbal-lab-org/make_prg v0.5.0", what="NodeFactory.build re-entry below a parent"),
                       cases=cases), fh, separators=(",", ":"))
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
