"""PRG string → little-endian uint32 stream (API of make_prg/utils/prg_encoder.py).
encode_array() is the array-at-a-time form the batch driver uses; encode()/_encode_unit() keep the reference's semantics
and errors unit by unit."""
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
        fast = self.encode_array(prg)
        if fast is not None:
            return fast.tolist()
        return self._encode_units(prg)

    def _encode_units(self, prg: str) -> PRG_Ints:
        """Unit by unit, as the reference (:44-91): also the path that raises its errors."""
        out: List[int] = []
        for unit in prg.split():
            out.extend(self._encode_unit(unit))
        return out

    def encode_array(self, prg: str):
        """The same stream as a uint32 array, array-at-a-time (a config-C PRG is ~10^5 characters).  Returns None when
        the text needs the unit-by-unit path: anything but space-separated pure-DNA / pure-digit units, an odd marker
        seen more than twice, or an encoder that already holds marker counts from an earlier call."""
        if self._site_entry_markers:
            return None
        if self.encoding is PrgEncoder.encoding:          # the default alphabet: libmprg's one-pass host encoder
            from . import native
            arr = native.prg_encode(prg)
            if arr is not None:
                odd = arr[(arr > 4) & (arr % 2 == 1)]
                self._site_entry_markers = {int(v): 2 for v in np.unique(odd)} if len(odd) else {}
                return arr
        try:
            b = np.frombuffer(prg.encode("ascii"), dtype=np.uint8)
        except UnicodeEncodeError:
            return None
        n = len(b)
        if n == 0:
            return np.zeros(0, np.uint32)
        lut = np.zeros(256, np.int64)
        for ch, v in self.encoding.items():
            if len(ch) != 1 or not (0 < int(v) < 2 ** 32):
                return None
            lut[ord(ch.upper())] = v
            lut[ord(ch.lower())] = v
        dna = lut[b]
        is_dna = dna > 0
        is_digit = (b >= 48) & (b <= 57)
        is_space = b == 32
        if not (is_dna | is_digit | is_space).all() or (is_dna & is_digit).any():
            return None
        tok = ~is_space
        start = tok & ~np.concatenate(([False], tok[:-1]))            # first character of every unit
        end = tok & ~np.concatenate((tok[1:], [False]))
        unit_id = np.cumsum(start) - 1
        n_units = int(start.sum())
        if n_units == 0:
            return np.zeros(0, np.uint32)
        digits_per_unit = np.bincount(unit_id[tok], weights=is_digit[tok], minlength=n_units).astype(np.int64)
        len_per_unit = np.bincount(unit_id[tok], minlength=n_units)
        numeric = digits_per_unit == len_per_unit
        if ((digits_per_unit > 0) & ~numeric).any() or (len_per_unit[numeric] > 18).any():
            return None                                                 # mixed unit (EncodeError) or absurdly long number
        # value of every numeric unit: digits weighted by their distance to the unit's end
        s_pos, e_pos = np.nonzero(start)[0], np.nonzero(end)[0]
        val = np.zeros(n_units, np.int64)
        dpos = np.nonzero(is_digit)[0]
        du = unit_id[dpos]
        np.add.at(val, du, (b[dpos].astype(np.int64) - 48) * 10 ** (e_pos[du] - dpos))
        num_units = np.nonzero(numeric)[0]
        mv = val[num_units]
        odd = (mv % 2) == 1
        if odd.any():                                                   # 1st occurrence stays, 2nd becomes the even marker
            ov = mv[odd]
            order = np.argsort(ov, kind="stable")
            sv = ov[order]
            first = np.concatenate(([True], sv[1:] != sv[:-1]))
            grp_start = np.maximum.accumulate(np.where(first, np.arange(len(sv)), 0))
            rank = np.arange(len(sv)) - grp_start
            if (rank > 1).any():
                return None                                             # the unit path raises the reference's ValueError
            bump = np.zeros(len(ov), np.int64)
            bump[order] = rank
            mv = mv.copy()
            mv[np.nonzero(odd)[0]] += bump
            uniq, cnt = np.unique(ov, return_counts=True)
            self._site_entry_markers = {int(u): int(c) for u, c in zip(uniq, cnt)}
        if (mv >= 2 ** 32).any():
            return None
        # one output per DNA character, one per numeric unit (at its first character), in text order
        emit = is_dna | (start & numeric[np.maximum(unit_id, 0)] & tok)
        out = np.where(is_dna, dna, 0)
        out[s_pos[num_units]] = mv
        return out[emit].astype(np.uint32)

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
