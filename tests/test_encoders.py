"""Host-side output encoders: the array-at-a-time .bin encoder must equal the unit-by-unit one (reference
utils/prg_encoder.py:44-91) on every golden PRG and leave the error cases to it."""
import numpy as np
import pytest

from make_prg_amd.utils.prg_encoder import EncodeError, PrgEncoder


def test_array_encoder_equals_unit_encoder_on_goldens(golden_integration, golden_synthetic):
    prgs = [l["expect"]["prg"] for c in golden_integration["cases"] for l in c["loci"] if "prg" in l.get("expect", {})]
    prgs += [l["expect"]["prg"] for l in golden_synthetic["loci"] if "prg" in l.get("expect", {})]
    assert len(prgs) > 40
    for prg in prgs:
        fast = PrgEncoder().encode_array(prg)
        assert fast is not None and fast.dtype == np.uint32
        assert fast.tolist() == PrgEncoder()._encode_units(prg)


@pytest.mark.parametrize("text,expect", [("", []), ("ACGT", [1, 2, 3, 4]), (" 5 A 6 C 5 ", [5, 1, 6, 2, 6]),
                                         ("a 5 g 6 T 5 cc 7 A 8 C 7 ", [1, 5, 3, 6, 4, 6, 2, 2, 7, 1, 8, 2, 8]), ("12", [12])])
def test_small_cases(text, expect):
    assert PrgEncoder().encode(text) == expect
    assert PrgEncoder().encode_array(text).tolist() == expect


def test_errors_come_from_the_unit_path():
    assert PrgEncoder().encode_array(" 5 A 6 C 5  5 A 6 T 5 ") is None        # odd marker a third time
    with pytest.raises(ValueError):
        PrgEncoder().encode(" 5 A 6 C 5  5 A 6 T 5 ")
    assert PrgEncoder().encode_array("AC5G") is None                           # mixed unit
    with pytest.raises(EncodeError):
        PrgEncoder().encode("AC5G")
    enc = PrgEncoder()
    assert enc.encode(" 5 A 6 C 5 ") == [5, 1, 6, 2, 6]
    with pytest.raises(ValueError):                                             # marker counts persist in the encoder
        enc.encode(" 5 A")


# ---- libmprg's one-pass host encoders (include/mprg.h: mprg_prg_encode_host, mprg_gfa_text_host)
@pytest.fixture(scope="module")
def native_lib():
    """The emulation build exports the same host functions as the HIP build (same sources)."""
    import ctypes
    from make_prg_amd.backend import bind
    from make_prg_amd.utils import native
    from tests.emu.backend import build_emu
    native.set_library(bind(ctypes.CDLL(build_emu())))
    yield native
    native.set_library(None)
    native._tried = False


def test_native_encoders_on_goldens(native_lib, golden_integration, golden_synthetic):
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.gfa import GFA_Output, gfa_text_single_pass
    prgs = [l["expect"]["prg"] for c in golden_integration["cases"] for l in c["loci"] if "prg" in l.get("expect", {})]
    prgs += [l["expect"]["prg"] for l in golden_synthetic["loci"] if "prg" in l.get("expect", {})]
    n_gfa = 0
    for prg in prgs:
        arr = native_lib.prg_encode(prg)
        assert arr is not None and arr.tolist() == PrgEncoder()._encode_units(prg)
        want = orc.gfa_text(prg)
        assert gfa_text_single_pass(prg) == want                   # the Python single pass
        got = native_lib.gfa_text(prg)
        assert got is not None and got.decode() == want            # the C single pass
        assert GFA_Output.gfa_text(prg) == want and GFA_Output.gfa_bytes(prg) == want.encode()
        n_gfa += 1
    assert n_gfa > 40


def test_native_encoders_leave_odd_strings_to_the_reference_shaped_path(native_lib):
    from make_prg_amd.utils.gfa import GFA_Output
    for text in (" 5 A 6 C 5  5 A 6 T 5 ", "AC5G", "A 7 C 8 G 7 T", "A 5 C 6 G", "AC 5 G 5 T"):
        assert native_lib.gfa_text(text) is None
    for text in (" 5 A 6 C 5  5 A 6 T 5 ", "AC5G", "AXG"):
        assert native_lib.prg_encode(text) is None
    with pytest.raises(ValueError):
        PrgEncoder().encode(" 5 A 6 C 5  5 A 6 T 5 ")
    with pytest.raises(AssertionError):                            # the reference's own assertion (utils/gfa.py:49-52)
        GFA_Output.gfa_text("A 5 C 6 G")
    # tiny marker-dense strings need more than 3 bytes of GFA per byte of PRG: the wrapper retries with the bound
    prg = "A 5  6  6  5 " * 1
    assert native_lib.gfa_text(prg).decode() == GFA_Output.gfa_text(prg)


def test_native_fasta_parser_equals_the_python_parser(native_lib):
    """mprg_fasta_scan_host / mprg_fasta_fill_host against read_fasta_alignment (the Bio.AlignIO-shaped parser) on wrapped
    lines, CRLF, blanks inside sequences, lower case, text before the first record, empty input, ragged rows, and the
    inputs the native parser hands back (form feed, old-Mac line ends, non-ASCII)."""
    import numpy as np
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils import native
    lib = native._lib
    cases = [">a desc here  \nACGT\nac gt\n>b\r\nAC\tGT\r\nACGT\r\n", "junk\n>x\nAC\n\n>y \nGT\n", ">only\n", "", "no records\n",
             ">a\nACG\n>b\nAC\n", ">a\nAC\x0cGT\n>b\nACGT\n", ">é\nAC\n", ">a\rAC\r>b\rGT\r", ">a\nacgtn\n>b\nACGTN",
             ">a\n\n\nAC\n>b\nA\nC", ">a  two words\t\nAC-N\n>a  two words\nRYKM"]
    try:
        for text in cases:
            res = []
            for use in (True, False):
                native.set_library(lib if use else None)
                try:
                    m = load_alignment_text(text)
                    res.append((m.data.tobytes(), m.data.shape, m.ids, m.descriptions))
                except ValueError as e:
                    res.append(("ValueError", str(e)))
            assert res[0] == res[1], repr(text)
        native.set_library(lib)
        assert native.parse_fasta(cases[0]) is not None and native.parse_fasta(cases[6]) is None
    finally:
        native.set_library(lib)
