"""Container-only: run the REAL reference (under oracle/refshim) on its own integration inputs and check
that the pinned oracle configuration (n_init=10, OMP=1, OPENBLAS_CORETYPE=Haswell) reproduces every committed
truth .prg.fa. Usage: python -m oracle.tools.check_reference_truth"""
import sys, os, re, zipfile, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.refshim.bootstrap as rb
rb.preset_env()
rb.install()
from pathlib import Path
from make_prg.prg_builder import PrgBuilder
from make_prg.utils.seq_utils import SequenceCurationError
from make_prg.utils.input_output_files import InputOutputFilesFromMSA

D = Path("/root/reference/tests/integration_tests/data")
T = D / "truth_output"


def build(path, N=5, L=7):
    locus = InputOutputFilesFromMSA.remove_known_input_extensions(path.name)
    try:
        b = PrgBuilder(locus, path, "fasta", N, L)
        return locus, b.build_prg()
    except SequenceCurationError:
        return locus, None


def truth_prgs(case):
    fa = T / case / f"{case}.prg.fa"
    out = {}
    if not fa.exists():
        return out
    lines = fa.read_text().split("\n")
    for i in range(0, len(lines) - 1, 2):
        out[lines[i][1:]] = lines[i + 1]
    return out


ok = bad = 0
for case in sorted(os.listdir(T)):
    src = D / case
    if (D / f"{case}.fa").exists():
        files = [D / f"{case}.fa"]
    elif src.is_dir():
        files = sorted(p for p in src.iterdir() if p.is_file())
    elif case == "match_compressed":
        files = [D / "match.fa.gz"]
    elif case == "match_overwrite":
        files = [D / "match.fa"]
    else:
        print("skip", case); continue
    truth = truth_prgs(case)
    kw = {}
    for f in files:
        locus, prg = build(f, **kw)
        t = truth.get(locus if len(files) > 1 else case)
        if t is None and prg is None:
            continue
        if prg == t:
            ok += 1
        else:
            bad += 1
            print("MISMATCH", case, locus, (prg or "")[:60], "...", (t or "")[:60])
print("ok", ok, "bad", bad)
