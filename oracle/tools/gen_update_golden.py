"""Container-only: golden vectors for BASELINE.json config 5 — `make_prg update` — produced by the REAL reference.

For each of the reference's own update integration cases (tests/integration_tests/test_update.py) the unmodified
reference (imported from /root/reference under oracle/refshim) builds the base locus/loci with `from_msa`, then runs
`update` on the committed denovo_paths.txt with the REAL bundled MAFFT v7.490; every MAFFT call is recorded
(previous_msa.fa text + new_sequences.fa text -> updated_msa.fa text) so that the GPU box, which has no MAFFT, can
replay `LeafNode._update_leaf` through make_prg_amd's ReplayAligner.  The run is cross-checked against the reference's
committed truth files (truth_output_update/<case>/: .prg.fa, .prg.bin[.zip], .prg.gfa[.zip] byte-identical) before
anything is written, and Bio.pairwise2 (absent here; stand-in oracle/refshim/Bio/pairwise2.py) is checked against the
reference's TestAlign known answers.  Recorded per case: inputs (FASTA texts, denovo_paths.txt text, flags), the
aligner replay table, every `align(ref, alt)` call the denovo parser made, and per locus the expected PRG, .bin / .gfa
hashes, full recursion-tree dump (node ids, kinds, nesting levels, per-node alignment rows), prg_index, counters, stats.

    python -m oracle.tools.gen_update_golden
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.refshim.bootstrap as rb

rb.preset_env()
rb.install()

import gzip
import hashlib
import json
import shutil
import tempfile
import zipfile
from argparse import Namespace
from pathlib import Path

import make_prg.update.denovo_variants as dv
from make_prg.prg_builder import PrgBuilderZipDatabase
from make_prg.recursion_tree import LeafNode, MultiClusterNode
from make_prg.subcommands import from_msa, output_type, update
from make_prg.utils import msa_aligner
from make_prg.utils.seq_utils import align

DATA = Path("/root/reference/tests/integration_tests/data")
OUT = Path(ROOT) / "tests" / "golden" / "update.json.gz"

# (update case, base input relative to DATA, -L of the base build, -D long deletion threshold, output types)
CASES = [
    ("match_update_simple", "match.fa", 7, 1000000, "a"),
    ("match_update_simple_with_long_deletion", "match.fa", 7, 10, "a"),
    ("match_update_complex", "match.fa", 7, 1000000, "a"),
    ("match.nonmatch.match_update", "match.nonmatch.match.fa", 7, 1000000, "a"),
    ("nested_snps_seq_backgrounds_update", "nested_snps_seq_backgrounds.fa", 3, 1000000, "a"),
    ("strict_insertions_and_deletions_update", "match.fa", 7, 1000000, "a"),
    ("match_update_simple_prg_only", "match.fa", 7, 1000000, "p"),
    ("match_update_simple_gfa_only", "match.fa", 7, 1000000, "g"),
    ("match_update_simple_bin_only", "match.fa", 7, 1000000, "b"),
    ("sample_example_update", "sample_example", 7, 1000000, "a"),
]
ALIGN_KNOWN = [("TA", "TGA", ("T-A", "TGA")), ("TTTAAA", "TTTGGAAA", ("TTT--AAA", "TTTGGAAA")), ("CA", "TGA", ("C-A", "TGA")),
               ("AAACCCGGGTTT", "GTGAAAGGCCCTATAGGGAAATTTAA", ("---AAA--CCC----GGG---TTT--", "GTGAAAGGCCCTATAGGGAAATTTAA")),
               ("AAAAACCCCCGGGGGTTTTT", "GTGAATAAGGCCGCCTATAGGCGGAAATTATTAA",
                ("---AAAAA--CCCCC----GGGGG---TTTTT--", "GTGAATAAGGCCGCCTATAGGCGGAAATTATTAA")),
               ("", "ACGT", ("----", "ACGT")), ("ACGT", "", ("ACGT", "----")), ("", "", ("", ""))]


def sha(b):
    return hashlib.sha256(b if isinstance(b, bytes) else b.encode()).hexdigest()


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


def zip_members(path):
    with zipfile.ZipFile(path) as z:
        return {n: z.read(n) for n in sorted(z.namelist())}


def outputs_of(prefix: str):
    """{kind: bytes or {member: bytes}} of whatever a run wrote under `prefix`."""
    out = {}
    for ext in (".prg.fa", ".prg.bin", ".prg.gfa"):
        if os.path.exists(prefix + ext):
            out[ext] = open(prefix + ext, "rb").read()
    for ext in (".prg.bin.zip", ".prg.gfa.zip"):
        if os.path.exists(prefix + ext):
            out[ext] = zip_members(prefix + ext)
    return out


def check_against_truth(case, got):
    truth = outputs_of(str(DATA / "truth_output_update" / case / case))
    assert set(truth) == set(got), (case, sorted(truth), sorted(got))
    for k in truth:
        assert truth[k] == got[k], f"{case}: {k} differs from the reference's committed truth"
    return sorted(truth)


def run_case(spec, tmp, rec_dir, align_calls):
    case, base, L, threshold, otype = spec
    for f in rec_dir.iterdir():
        f.unlink()
    del align_calls[:]
    base_path = DATA / base
    files = sorted(base_path.iterdir()) if base_path.is_dir() else [base_path]
    base_prefix = str(tmp / case / "base" / "base")
    from_msa.run(Namespace(input=str(base_path), suffix="", output_prefix=base_prefix, alignment_format="fasta",
                           log=None, max_nesting=5, min_match_length=L, output_type=output_type.OutputType("a"),
                           force=True, threads=1, verbose=False))
    upd_prefix = str(tmp / case / "out" / case)
    denovo = DATA / case / "denovo_paths.txt"
    update.run(Namespace(denovo_paths=str(denovo), update_DS=Path(base_prefix + ".update_DS.zip"),
                         output_prefix=upd_prefix, long_deletion_threshold=threshold, log=None,
                         output_type=output_type.OutputType(otype), force=True, threads=1, verbose=False))
    got = outputs_of(upd_prefix)
    checked = check_against_truth(case, got)
    loci = {}
    if otype in ("a", "p"):
        db = PrgBuilderZipDatabase(Path(upd_prefix + ".update_DS.zip"))
        db.load()
        for locus in db.get_loci_names():
            b = db.get_PrgBuilder(locus)
            prg = b.build_prg()
            loci[locus] = dict(prg=prg, tree=tree_dump(b.root), next_node_id=b.next_node_id, site_num=b.site_num,
                               prg_index=sorted([s, e, n.node_id] for (s, e), n in b.prg_index.items()))
        db.close()
    files_expect = {}
    for k, v in got.items():
        files_expect[k] = {m: sha(x) for m, x in v.items()} if isinstance(v, dict) else sha(v)
    replay = [json.load(open(rec_dir / f)) for f in sorted(os.listdir(rec_dir))]
    return dict(case=case, N=5, L=L, long_deletion_threshold=threshold, output_type=otype,
                      inputs=[dict(name=f.name, fasta=f.read_text()) for f in files],
                      denovo_paths=denovo.read_text(), aligner_replay=replay, align_calls=list(align_calls),
                      expect=dict(files_sha256=files_expect, prg_fa=got.get(".prg.fa", b"").decode(), loci=loci),
                      checked_against_truth=checked)


def main():
    for a, b, want in ALIGN_KNOWN:          # tests/utils/test_seq_utils.py::TestAlign
        assert align(a, b) == want, (a, b, align(a, b))
    assert align("A", "T", 0, 0, 0, 0) == ("A-", "-T")
    tmp = Path(tempfile.mkdtemp(prefix="mprg_update_golden_"))
    rec_dir = tmp / "mafft_calls"
    rec_dir.mkdir()
    orig_mafft = msa_aligner.MAFFT.get_updated_alignment

    def recording(self, current_alignment, new_sequences):
        from io import StringIO
        from Bio import SeqIO
        buf = StringIO()
        SeqIO.write(current_alignment, buf, "fasta")
        updated = orig_mafft(self, current_alignment, new_sequences)
        n = len(os.listdir(rec_dir))
        with open(rec_dir / f"{os.getpid()}_{n}.json", "w") as fh:
            json.dump(dict(previous_msa=buf.getvalue(), new_sequences=sorted(new_sequences),
                           updated_rows=[[r.id, r.description, str(r.seq)] for r in updated]), fh)
        return updated

    msa_aligner.MAFFT.get_updated_alignment = recording
    align_calls = []
    orig_align = dv.align

    def recording_align(ref, alt, *a, **k):
        res = orig_align(ref, alt, *a, **k)
        align_calls.append([ref, alt, list(res)])
        return res

    dv.align = recording_align
    import multiprocessing as mp
    cases = []
    for spec in CASES:          # one forked child per case: the reference keeps its aligner and variants in a process-wide
        recv, send = mp.Pipe(False)   # singleton (update_shared_data.py), which is why its own tests run forked
        child = mp.get_context("fork").Process(target=lambda: send.send(run_case(spec, tmp, rec_dir, align_calls)))
        child.start()
        rec = recv.recv()
        child.join()
        cases.append(rec)
        print(rec["case"], "ok:", rec["checked_against_truth"], "| mafft calls", len(rec["aligner_replay"]), "| align calls",
              len(rec["align_calls"]), "| loci", len(rec["expect"]["loci"]))
    meta = dict(reference="iqbal-lab-org/make_prg v0.5.0 (unmodified, /root/reference, oracle/refshim)",
                mafft="bundled v7.490 (make_prg/utils/mafft-linux64), --auto --quiet --thread 1 --add",
                pinned=dict(n_init=10, OMP_NUM_THREADS=1, OPENBLAS_CORETYPE=rb.PINNED_CORETYPE),
                align_known_answers=[[a, b, list(w)] for a, b, w in ALIGN_KNOWN])
    with gzip.open(OUT, "wt") as fh:
        json.dump(dict(meta=meta, cases=cases), fh, separators=(",", ":"))
    print("wrote", OUT, OUT.stat().st_size, "bytes")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
