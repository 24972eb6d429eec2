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
    """reference utils/seq_utils.py:193-216, on the device (mprg_column_masks + mprg_compact_columns).  Bytes outside the
    kernels' alphabet (the reference drops all-gap columns of any text) and lower case (the cell codes fold it) keep the
    host's own column selection."""
    if len(alignment) == 0 or alignment.get_alignment_length() == 0:
        return alignment
    codes = encode(alignment.data)
    if (codes == 255).any():
        keep = ~(alignment.data == ord(GAP)).all(axis=0)
        return MSA(_data=alignment.data[:, keep], _ids=alignment.ids, _descs=alignment.descriptions)
    if not np.array_equal(decode(codes), alignment.data):
        keep = _masks_of(alignment) != _eng.BIT_GAP
        return MSA(_data=alignment.data[:, keep], _ids=alignment.ids, _descs=alignment.descriptions)
    dense = _eng.BatchEngine(get_backend(), 1, 1).compact_columns(alignment)
    return MSA(_data=decode(dense), _ids=alignment.ids, _descs=alignment.descriptions)


def has_empty_sequence(alignment: MSA, interval: Tuple[int, int]) -> bool:
    """reference utils/seq_utils.py:37-42."""
    sub = alignment.data[:, interval[0]:interval[1] + 1]
    return bool(len(alignment)) and bool((sub == ord(GAP)).all(axis=1).any())


def get_number_of_unique_ungapped_sequences(sub_alignment: MSA) -> int:
    return _eng.BatchEngine(get_backend(), 1, 1).row_groups(sub_alignment)[0]


def get_number_of_unique_gapped_sequences(sub_alignment: MSA) -> int:
    return _eng.BatchEngine(get_backend(), 1, 1).row_groups(sub_alignment)[1]


# ---------------------------------------------------------------------------------------------------- pairwise alignment
def _affine_gap(length: int, open_score: float, extend_score: float) -> float:
    return 0 if length <= 0 else open_score + extend_score * (length - 1)


def _score_key(x: float) -> int:
    """Scores are compared after rounding to 1/1000 (Biopython pairwise2's rint)."""
    return int(x * 1000 + 0.5)


def align(seq1: str, seq2: str, match_score: float = 2, mismatch_score: float = -0.9, gap_open_score: float = -1.1,
          gap_extend_score: float = -1) -> Tuple[str, str]:
    """Global alignment of two short sequences with affine gaps (reference utils/seq_utils.py:161-190, which takes
    the FIRST alignment of Bio.pairwise2.align.globalms(..., one_alignment_only=True)).  Used by `update` to split a
    denovo variant that spans several leaves (update/denovo_variants.py).

    Which of the equally good alignments comes first is part of the contract, so this follows pairwise2's published
    procedure: forward pass with three running scores per cell (pair, gap in seq1, gap in seq2; end gaps penalised;
    opening a gap costs `gap_open_score`, each further column `gap_extend_score`), every way a cell's best score can be
    reached remembered as a bit set {1: open gap in seq1, 2: pair, 4: open gap in seq2, 8: extend gap in seq1,
    16: extend gap in seq2}; backward pass depth-first, trying the bits of a cell in that order, with the branches not
    taken kept on a stack, never letting a gap in seq1 directly precede (in traceback order) a gap in seq2.
    Known answers: tests/golden/update.json.gz `align_known_answers` (the reference's own TestAlign cases)."""
    n1, n2 = len(seq1), len(seq2)
    if n1 == 0:
        return GAP * n2, seq2
    if n2 == 0:
        return seq1, GAP * n1
    go, ge = gap_open_score, gap_extend_score
    score = [[0.0] * (n2 + 1) for _ in range(n1 + 1)]
    ways = [[0] * (n2 + 1) for _ in range(n1 + 1)]          # 0 on the borders: "finish with end gaps"
    for i in range(n1 + 1):
        score[i][0] = _affine_gap(i, go, ge)
    for j in range(n2 + 1):
        score[0][j] = _affine_gap(j, go, ge)
    gap2 = [0.0] + [_affine_gap(j, 2 * go, ge) for j in range(1, n2 + 1)]    # best score ending with a gap in seq2, per column
    for i in range(1, n1 + 1):
        gap1 = _affine_gap(i, 2 * go, ge)                                    # best score ending with a gap in seq1, this row
        a = seq1[i - 1]
        up, cur, wr = score[i - 1], score[i], ways[i]
        for j in range(1, n2 + 1):
            pair = up[j - 1] + (match_score if a == seq2[j - 1] else mismatch_score)
            g1_open, g1_ext = cur[j - 1] + go, gap1 + ge
            gap1 = max(g1_open, g1_ext)
            g2_open, g2_ext = up[j] + go, gap2[j] + ge
            gap2[j] = max(g2_open, g2_ext)
            best = max(pair, gap2[j], gap1)
            cur[j] = best
            kb, k1, k2 = _score_key(best), _score_key(gap1), _score_key(gap2[j])
            w = 2 if _score_key(pair) == kb else 0
            if k1 == kb:
                w += (1 if _score_key(g1_open) == k1 else 0) + (8 if _score_key(g1_ext) == k1 else 0)
            if k2 == kb:
                w += (4 if _score_key(g2_open) == k2 else 0) + (16 if _score_key(g2_ext) == k2 else 0)
            wr[j] = w

    def walk_gap(out1, out2, i, j, after_gap2, in_seq1, todo):
        """An extended gap: step back through it; every point where it could have been opened is a branch for later."""
        target = _score_key(score[i][j])
        dead = False
        for n in range(j if in_seq1 else i):
            if in_seq1:
                j -= 1
                out1.append(GAP); out2.append(seq2[j])
            else:
                i -= 1
                out1.append(seq1[i]); out2.append(GAP)
            if _score_key(score[i][j] + _affine_gap(n + 1, go, ge)) == target and n > 0:
                if not ways[i][j]:
                    break
                todo.append((out1[:], out2[:], i, j, after_gap2, ways[i][j]))
            if not ways[i][j]:
                dead = True
        return i, j, dead

    todo = [([], [], n1, n2, False, ways[n1][n2])]
    while todo:
        out1, out2, i, j, after_gap2, w = todo.pop()
        dead = False
        while (i > 0 or j > 0) and not dead:
            here = (out1[:], out2[:], i, j, after_gap2)
            if not w:                                   # a border: the rest is end gaps
                if j and after_gap2:
                    dead = True
                else:
                    out1.extend(reversed(seq1[:i])); out2.extend(reversed(seq2[:j]))
                    if i > j:
                        out2.extend(GAP * (len(out1) - len(out2)))
                    elif j > i:
                        out1.extend(GAP * (len(out2) - len(out1)))
                break
            if w & 1:
                w -= 1
                if after_gap2:
                    dead = True
                else:
                    j -= 1
                    out1.append(GAP); out2.append(seq2[j])
            elif w & 2:
                w -= 2
                i -= 1; j -= 1
                out1.append(seq1[i]); out2.append(seq2[j])
                after_gap2 = False
            elif w & 4:
                w -= 4
                i -= 1
                out1.append(seq1[i]); out2.append(GAP)
                after_gap2 = True
            elif w in (8, 24):
                w -= 8
                if after_gap2:
                    dead = True
                else:
                    i, j, dead = walk_gap(out1, out2, i, j, False, True, todo)
            elif w == 16:
                w -= 16
                after_gap2 = True
                i, j, dead = walk_gap(out1, out2, i, j, True, False, todo)
            if w:
                todo.append(here + (w,))
            w = ways[i][j]
        if not dead:
            return "".join(reversed(out1)), "".join(reversed(out2))
    raise RuntimeError("align(): no traceback found")
