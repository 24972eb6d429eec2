"""PRG string → little-endian uint32 stream (API of make_prg/utils/prg_encoder.py).  Vectorised host code."""
from typing import BinaryIO, Dict, List

import numpy as np


class ConversionError(Exception):
    pass


class EncodeError(Exception):
    pass


PRG_Ints = List[int]
BYTES_PER_INT = 4
ENDIANNESS = "little"


def to_bytes(integer: int):
    return integer.to_bytes(BYTES_PER_INT, ENDIANNESS)


class PrgEncoder:
    """A C G T → 1 2 3 4; the second occurrence of an odd site marker becomes the even marker (reference :22-91)."""
    encoding = {"A": 1, "C": 2, "G": 3, "T": 4}

    def __init__(self, encoding: Dict[str, int] = None):
        if encoding is not None:
            self.encoding = encoding
        self._site_entry_markers: Dict[int, int] = {}

    def encode(self, prg: str) -> PRG_Ints:
        out: List[int] = []
        for unit in prg.split():
            out.extend(self._encode_unit(unit))
        return out

    @staticmethod
    def write(encoding: List[int], ostream: BinaryIO):
        ostream.write(np.asarray(encoding, dtype="<u4").tobytes())

    def _dna_to_int(self, input_char: str) -> int:
        c = input_char.upper()
        if c not in self.encoding:
            raise ConversionError(f"Char '{c}' is not in {self.encoding}")
        return self.encoding[c]

    def _encode_unit(self, unit: str) -> List[int]:
        if not unit:
            raise EncodeError("Cannot encode an empty string")
        if all(c.upper() in self.encoding for c in unit):
            return [self._dna_to_int(c) for c in unit]
        if unit.isdigit():
            marker = int(unit)
            if marker % 2 == 0:
                return [marker]
            seen = self._site_entry_markers.get(marker, 0) + 1
            if seen > 2:
                raise ValueError(f"Prg error: odd site marker {marker} found >2 times")
            self._site_entry_markers[marker] = seen
            return [marker] if seen == 1 else [marker + 1]
        raise EncodeError("Unit {} contains invalid characters".format(unit))
