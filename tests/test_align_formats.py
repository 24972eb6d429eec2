"""`-f`: the alignment formats other than FASTA (make_prg_amd/utils/align_formats.py; reference utils/io_utils.py:17-31 reads
any Biopython AlignIO format).  The readers are restated from the format definitions — Biopython is not available to pin them —
so they are checked against the FASTA reader on hand-written files of the same alignment, and through the command line."""
import numpy as np
import pytest

from make_prg_amd.msa import load_alignment_text
from make_prg_amd.utils.align_formats import FORMATS, read_alignment

FASTA = ">s1 first\nACGT-ACGTTTA\n>s2\nACGTTACGTTCA\n>s3\nAC-TTACGNTCA\n"
FILES = {
    "clustal": "CLUSTAL W (1.83) multiple sequence alignment\n\n\ns1   ACGT-A 5\ns2   ACGTTA 6\ns3   AC-TTA 5\n     ** * *\n\n"
               "s1   CGTTTA 11\ns2   CGTTCA 12\ns3   CGNTCA 11\n     ** * *\n",
    "stockholm": "# STOCKHOLM 1.0\n#=GF ID test\ns1 ACGT-A\ns2 ACGTTA\ns3 AC-TTA\n#=GC SS_cons ......\n\ns1 CGTTTA\ns2 CGTTCA\ns3 CGNTCA\n//\n",
    "phylip": " 3 12\ns1        ACGT-A CGT\ns2        ACGTTA CGT\ns3        AC-TTA CGN\n\nTTA\nTCA\nTCA\n",
    "phylip-sequential": " 3 12\ns1        ACGT-A\nCGTTTA\ns2        ACGTTACGTTCA\ns3        AC-TTA\nCGN\nTCA\n",
    "phylip-relaxed": " 3 12\nsample_one_long_name ACGT-ACGTTTA\ns2 ACGTTACGTTCA\ns3 AC-TTACGNTCA\n",
}


def test_every_format_reads_the_same_alignment():
    want = load_alignment_text(FASTA)
    assert sorted(FILES) == sorted(FORMATS)
    for fmt, text in FILES.items():
        got = load_alignment_text(text, False, fmt)
        assert np.array_equal(got.data, want.data), fmt          # upper-cased, N replaced by the seeded column consensus
        assert got.ids[1:] == ["s2", "s3"] and got.ids[0] == ("sample_one_long_name" if fmt == "phylip-relaxed" else "s1")
        assert got.descriptions == got.ids                       # these formats carry no free-text description


def test_errors():
    with pytest.raises(ValueError, match="not supported"):
        read_alignment(FASTA, "nexus")
    with pytest.raises(ValueError, match="No records found in handle"):
        read_alignment("\n\n", "clustal")
    with pytest.raises(ValueError, match="More than one record"):
        read_alignment(FILES["stockholm"] + FILES["stockholm"], "stockholm")
    with pytest.raises(ValueError, match="same length"):
        read_alignment("# STOCKHOLM 1.0\na ACGT\nb ACG\n//\n", "stockholm")
    with pytest.raises(ValueError, match="known CLUSTAL header"):
        read_alignment("FOO 1.0\n\na ACGT\n", "clustal")
    with pytest.raises(ValueError, match="out of order"):
        read_alignment("CLUSTAL X\n\na ACGT\nb ACGT\n\nb ACGT\na ACGT\n", "clustal")
    with pytest.raises(ValueError, match="two integers"):
        read_alignment("a ACGT\n", "phylip")
    with pytest.raises(ValueError, match="sites of the header"):
        read_alignment(" 2 5\na         ACGT\nb         ACGT\n", "phylip")


def test_command_line_with_a_clustal_directory(tmp_path, golden_integration):
    """from_msa -f clustal on CLUSTAL renderings of reference test alignments: the same PRGs as from their FASTA files."""
    from make_prg_amd import __main__ as cli, device
    from make_prg_amd.msa import load_alignment_text as load
    from tests.emu.backend import EmuBackend
    device.set_backend(EmuBackend())
    try:
        d_fa, d_cl = tmp_path / "fa", tmp_path / "cl"
        d_fa.mkdir(); d_cl.mkdir()
        n = 0
        for case in golden_integration["cases"]:
            if case["case"] in ("several", "match.nonmatch") and (case["N"], case["L"]) == (5, 7):
                for l in case["loci"]:
                    msa = load(l["fasta"])
                    name = l["file"].replace(".gz", "").replace(".fa", "")
                    (d_fa / (name + ".fa")).write_text(l["fasta"])
                    rows = [(i, r.tobytes().decode()) for i, r in zip(msa.ids, msa.data)]
                    blocks = []
                    for lo in range(0, len(rows[0][1]), 50):          # interleaved blocks of 50 columns
                        blocks.append("".join(f"{i:<20s} {s[lo:lo + 50]}\n" for i, s in rows) + "\n")
                    (d_cl / (name + ".aln")).write_text("CLUSTAL W (1.83) multiple sequence alignment\n\n\n" + "\n".join(blocks))
                    n += 1
        assert n >= 3
        cli.main(["from_msa", "-i", str(d_fa), "-o", str(tmp_path / "o_fa" / "x"), "-O", "p"])
        cli.main(["from_msa", "-i", str(d_cl), "-o", str(tmp_path / "o_cl" / "x"), "-O", "p", "-f", "clustal", "--suffix", ".aln"])

        def prgs(path):          # {locus: PRG}; ".aln" is not one of the extensions the reference strips from a locus name
            lines = path.read_text().splitlines()
            return {name[1:].replace(".aln", ""): prg for name, prg in zip(lines[0::2], lines[1::2])}

        got, want = prgs(tmp_path / "o_cl" / "x.prg.fa"), prgs(tmp_path / "o_fa" / "x.prg.fa")
        assert got == want and len(want) >= 3
    finally:
        device.set_backend(None)
