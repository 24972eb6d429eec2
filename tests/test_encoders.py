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
