"""Container-only: the oracle against the REAL reference (imported from /root/reference under oracle/refshim, pinned
KMeans configuration) on seeded random alignments — the generators the large GPU parity sweeps use
(tests/random_msas.py and the medium generator of tools/parity_sweep_nasty.py).  Compares PRG, next_node_id and the
recursion tree, or the common SequenceCurationError.

    python -m oracle.tools.check_oracle_vs_reference_random [n_small] [n_medium]
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.refshim.bootstrap as rb
rb.preset_env()
rb.install()

import tempfile
from pathlib import Path
from make_prg.prg_builder import PrgBuilder
from make_prg.recursion_tree import LeafNode, MultiClusterNode
from make_prg.utils.seq_utils import SequenceCurationError
import oracle.from_msa_oracle as orc
from tests.random_msas import random_cases
sys.path.insert(0, os.path.join(ROOT, "tools"))
from parity_sweep_nasty import medium_cases, COMBOS


def ref_tree(root):
    out = []

    def rec(n):
        kind = "leaf" if isinstance(n, LeafNode) else ("cluster" if isinstance(n, MultiClusterNode) else "interval")
        out.append(dict(id=n.node_id, kind=kind, level=n.nesting_level, parent=None if n.parent is None else n.parent.node_id,
                        rows=[[r.id, str(r.seq)] for r in n.alignment], children=[c.node_id for c in n.children]))
        for c in n.children:
            rec(c)
    rec(root)
    return out


def main():
    n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n_medium = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    tmp = Path(tempfile.mkdtemp())
    bad = total = errors = 0
    for ci, (N, L) in enumerate(COMBOS):
        texts = random_cases(1000 + ci, n_small) + medium_cases(1000 + ci, n_medium)
        for i, t in enumerate(texts):
            p = tmp / f"c{ci}_{i}.fa"
            p.write_text(t)
            try:
                b = PrgBuilder("x", p, "fasta", N, L)
                ref = (b.build_prg(), b.next_node_id, ref_tree(b.root))
            except SequenceCurationError:
                ref = "SequenceCurationError"
            except ValueError as e:        # unparsable input (ragged / empty): both sides reject at ingest
                ref = "ValueError"
            try:
                prg, ob, root = orc.build_locus_from_text(t, N, L)
                mine = (prg, ob.next_node_id, orc.tree_dump(root))
            except orc.SequenceCurationError:
                mine = "SequenceCurationError"
            except ValueError:
                mine = "ValueError"
            total += 1
            errors += isinstance(ref, str)
            if ref != mine:
                bad += 1
                print("MISMATCH", (N, L), i, ref if isinstance(ref, str) else ref[0][:60], mine if isinstance(mine, str) else mine[0][:60])
            p.unlink()
        print(f"N={N} L={L}: {len(texts)} alignments checked, mismatches so far {bad}", flush=True)
    print(f"{total} alignments ({errors} rejected by both), mismatches {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
