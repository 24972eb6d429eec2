"""Container-only stand-in for Bio.pairwise2 (Biopython 1.79 is the reference's pin, poetry.lock; not installed here).

Only what the reference's `update` path calls: pairwise2.align.globalms(seqA, seqB, match, mismatch, open, extend,
one_alignment_only=True) -> [Alignment(seqA, seqB, score, begin, end)] (make_prg/utils/seq_utils.py:161-190).
Restates Biopython's published algorithm: the affine-gap score/trace matrices of `_make_score_matrix_fast` (trace bits:
1 open gap in seqA, 2 match/mismatch, 4 open gap in seqB, 8 extend gap in seqA, 16 extend gap in seqB; scores compared
after rint(x * 1000 + 0.5)) and the stack-driven traceback of `_recover_alignments` / `_find_gap_open`, whose order of
preference decides WHICH of the equally good alignments comes first.  Pinned by the reference's own known answers:
tests/utils/test_seq_utils.py::TestAlign (run under this shim by oracle/tools/gen_update_golden.py) and the committed
truth_output_update directories, whose multi-leaf variants go through this function.
"""
from collections import namedtuple

Alignment = namedtuple("Alignment", ("seqA", "seqB", "score", "start", "end"))
_PRECISION = 1000


def rint(x, precision=_PRECISION):
    return int(x * precision + 0.5)


def _affine(length, open_, extend, penalize_extend_when_opening=False):
    if length <= 0:
        return 0
    penalty = open_ + extend * length
    if not penalize_extend_when_opening:
        penalty -= extend
    return penalty


def _score_matrices(A, B, match, mismatch, open_A, extend_A, open_B, extend_B):
    first_A_gap = _affine(1, open_A, extend_A)
    first_B_gap = _affine(1, open_B, extend_B)
    lenA, lenB = len(A), len(B)
    score = [[None] * (lenB + 1) for _ in range(lenA + 1)]
    trace = [[None] * (lenB + 1) for _ in range(lenA + 1)]
    for i in range(lenA + 1):
        score[i][0] = _affine(i, open_B, extend_B)
    for i in range(lenB + 1):
        score[0][i] = _affine(i, open_A, extend_A)
    col_score = [0] + [_affine(i, 2 * open_B, extend_B) for i in range(1, lenB + 1)]
    for row in range(1, lenA + 1):
        row_score = _affine(row, 2 * open_A, extend_A)
        for col in range(1, lenB + 1):
            nogap = score[row - 1][col - 1] + (match if A[row - 1] == B[col - 1] else mismatch)
            row_open = score[row][col - 1] + first_A_gap
            row_extend = row_score + extend_A
            row_score = max(row_open, row_extend)
            col_open = score[row - 1][col] + first_B_gap
            col_extend = col_score[col] + extend_B
            col_score[col] = max(col_open, col_extend)
            best = max(nogap, col_score[col], row_score)
            score[row][col] = best
            row_r, col_r, best_r = rint(row_score), rint(col_score[col]), rint(best)
            row_trace = (1 if rint(row_open) == row_r else 0) + (8 if rint(row_extend) == row_r else 0)
            col_trace = (4 if rint(col_open) == col_r else 0) + (16 if rint(col_extend) == col_r else 0)
            t = 2 if rint(nogap) == best_r else 0
            if row_r == best_r:
                t += row_trace
            if col_r == best_r:
                t += col_trace
            trace[row][col] = t
    return score, trace


def _finish(A, B, aliA, aliB, row, col):
    if row:
        aliA += A[row - 1::-1]
    if col:
        aliB += B[col - 1::-1]
    if row > col:
        aliB += "-" * (len(aliA) - len(aliB))
    elif col > row:
        aliA += "-" * (len(aliB) - len(aliA))
    return aliA, aliB


def _find_gap_open(A, B, aliA, aliB, row, col, col_gap, score, trace, in_process, open_, extend, target, direction):
    dead_end = False
    target_score = score[row][col]
    for n in range(target):
        if direction == "col":
            col -= 1
            aliA += "-"
            aliB += B[col:col + 1]
        else:
            row -= 1
            aliA += A[row:row + 1]
            aliB += "-"
        actual = score[row][col] + _affine(n + 1, open_, extend)
        if rint(actual) == rint(target_score) and n > 0:
            if not trace[row][col]:
                break
            in_process.append((aliA, aliB, row, col, col_gap, trace[row][col]))
        if not trace[row][col]:
            dead_end = True
    return aliA, aliB, row, col, dead_end


def _first_alignment(A, B, score, trace, open_A, extend_A, open_B, extend_B):
    lenA, lenB = len(A), len(B)
    in_process = [("", "", lenA, lenB, False, trace[lenA][lenB])]
    while in_process:
        dead_end = False
        aliA, aliB, row, col, col_gap, t = in_process.pop()
        while (row > 0 or col > 0) and not dead_end:
            cache = (aliA, aliB, row, col, col_gap)
            if not t:
                if col and col_gap:
                    dead_end = True
                else:
                    aliA, aliB = _finish(A, B, aliA, aliB, row, col)
                break
            elif t % 2 == 1:            # open gap in seqA
                t -= 1
                if col_gap:
                    dead_end = True
                else:
                    col -= 1
                    aliA += "-"
                    aliB += B[col:col + 1]
                    col_gap = False
            elif t % 4 == 2:            # match / mismatch
                t -= 2
                row -= 1
                col -= 1
                aliA += A[row:row + 1]
                aliB += B[col:col + 1]
                col_gap = False
            elif t % 8 == 4:            # open gap in seqB
                t -= 4
                row -= 1
                aliA += A[row:row + 1]
                aliB += "-"
                col_gap = True
            elif t in (8, 24):          # extend gap in seqA
                t -= 8
                if col_gap:
                    dead_end = True
                else:
                    col_gap = False
                    aliA, aliB, row, col, dead_end = _find_gap_open(A, B, aliA, aliB, row, col, col_gap, score, trace,
                                                                    in_process, open_A, extend_A, col, "col")
            elif t == 16:               # extend gap in seqB
                t -= 16
                col_gap = True
                aliA, aliB, row, col, dead_end = _find_gap_open(A, B, aliA, aliB, row, col, col_gap, score, trace,
                                                                in_process, open_B, extend_B, row, "row")
            if t:                       # another path to follow later
                in_process.append(cache + (t,))
            t = trace[row][col]
        if not dead_end:
            return aliA[::-1], aliB[::-1]
    raise RuntimeError("pairwise2 stand-in: no traceback found")


class _Align:
    def globalms(self, seqA, seqB, match, mismatch, open_, extend, one_alignment_only=False, **kw):
        if kw:
            raise NotImplementedError(f"pairwise2 stand-in: unsupported arguments {sorted(kw)}")
        if not seqA or not seqB:
            return []
        score, trace = _score_matrices(seqA, seqB, match, mismatch, open_, extend, open_, extend)
        a, b = _first_alignment(seqA, seqB, score, trace, open_, extend, open_, extend)
        return [Alignment(a, b, score[len(seqA)][len(seqB)], 0, len(a))]


align = _Align()
