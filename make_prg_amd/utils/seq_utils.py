"""Sequence helpers of the from_msa path, same names as make_prg/utils/seq_utils.py; the array work runs on the GPU."""
from typing import List, Tuple

import numpy as np

from .. import engine as _eng
from ..device import get_backend
from ..engine import SequenceCurationError, expand_sequences  # noqa: F401
from ..msa import MSA, Record, decode, encode

NONMATCH = "*"
GAP = "-"
Sequence = str
Sequences = List[str]


def is_non_match(letter: str) -> bool:
    return letter == NONMATCH


def is_gap(letter: str) -> bool:
    return letter == GAP


def ungap(seq: str) -> str:
    return seq.replace(GAP, "")


def remove_duplicates(seqs):
    seen = set()
    for x in seqs:
        if x not in seen:
            seen.add(x)
            yield x


def get_alignment_seqs(alignment: MSA):
    yield from alignment.rows_as_strings()


class SequenceExpander:
    """reference utils/seq_utils.py:77-158 (host: the strings are short; the device supplies the distinct rows)."""
    iupac = dict(_eng.IUPAC)
    expandable_bases = set(iupac)
    allowed_bases = expandable_bases | {"N"}
    standard_bases = set("ACGT")
    ambiguous_bases = expandable_bases - standard_bases

    @classmethod
    def get_expanded_sequences(cls, sequences: List[str]) -> Sequences:
        return expand_sequences(list(sequences))

    @classmethod
    def get_expanded_sequences_from_MSA(cls, alignment: MSA) -> Sequences:
        return expand_sequences([ungap(s) for s in alignment.rows_as_strings()])


def _masks_of(alignment: MSA) -> np.ndarray:
    """Per-column symbol presence masks of a whole alignment, computed by k_column_masks."""
    eng = _eng.BatchEngine(get_backend(), 1, 1)
    return eng.column_masks(alignment)


def get_consensus_from_MSA(alignment: MSA) -> str:
    """reference utils/seq_utils.py:219-239, on the device (mprg_column_masks)."""
    if len(alignment) == 0 or alignment.get_alignment_length() == 0:
        return ""
    codes = encode(alignment.data)
    if (codes == 255).any():
        return _consensus_host_rare(alignment)
    m = _masks_of(alignment)
    return decode_consensus(m)


def decode_consensus(mask: np.ndarray) -> str:
    m = mask & ~np.uint32(_eng.BIT_N)
    single = (m != 0) & ((m & (m - 1)) == 0) & ((m & _eng.BITS_IUPAC) == 0) & (m != _eng.BIT_GAP)
    out = np.full(mask.shape[0], ord("*"), np.uint8)
    out[single] = decode(np.log2(m[single]).astype(np.uint8))
    return out.tobytes().decode()


def _consensus_host_rare(alignment: MSA) -> str:
    # Bytes outside ACGT-RYKMSWN have no device code (such loci end in SequenceCurationError); keep the
    # reference's answer for direct callers of this helper.
    out = []
    for c in range(alignment.get_alignment_length()):
        col = set(alignment.data[:, c].tobytes().decode()) - {"N"}
        out.append(NONMATCH if (len(col) != 1 or col & SequenceExpander.ambiguous_bases or col == {GAP}) else next(iter(col)))
    return "".join(out)


def remove_columns_full_of_gaps_from_MSA(alignment: MSA) -> MSA:
    """reference utils/seq_utils.py:193-216; the all-gap column mask comes from the device."""
    if len(alignment) == 0 or alignment.get_alignment_length() == 0:
        return alignment
    codes = encode(alignment.data)
    if (codes == 255).any():
        keep = ~(alignment.data == ord(GAP)).all(axis=0)
    else:
        keep = _masks_of(alignment) != _eng.BIT_GAP
    return MSA(_data=alignment.data[:, keep], _ids=alignment.ids, _descs=alignment.descriptions)


def has_empty_sequence(alignment: MSA, interval: Tuple[int, int]) -> bool:
    """reference utils/seq_utils.py:37-42."""
    sub = alignment.data[:, interval[0]:interval[1] + 1]
    return bool(len(alignment)) and bool((sub == ord(GAP)).all(axis=1).any())


def get_number_of_unique_ungapped_sequences(sub_alignment: MSA) -> int:
    return _eng.BatchEngine(get_backend(), 1, 1).row_groups(sub_alignment)[0]


def get_number_of_unique_gapped_sequences(sub_alignment: MSA) -> int:
    return _eng.BatchEngine(get_backend(), 1, 1).row_groups(sub_alignment)[1]
