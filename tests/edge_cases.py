"""Small hand-made alignments for the edge cases the reference's unit tests probe (empty-allele padding, short
match absorption, gaps-only differences, IUPAC codes, N replacement, single row, narrow alignments)."""


def fa(rows):
    return "".join(f">s{i}\n{r}\n" for i, r in enumerate(rows))


EDGE_FASTAS = [
    (5, 7, [
        fa(["ACGTACGTACGT"]),                                        # single sequence
        fa(["ACGT", "ACGA"]),                                        # narrower than min_match_length
        fa(["AAAAAAAAAATAAAAAAAAAA", "AAAAAAAAAACAAAAAAAAAA", "AAAAAAAAAA-AAAAAAAAAA"]),   # empty allele padding
        fa(["AAAAAAAAAAT-CCCCCCCCCC", "AAAAAAAAAA-TCCCCCCCCCC"[:22], "AAAAAAAAAAT-CCCCCCCCCC"]),  # gap-only difference
        fa(["AAAAAAAAAARAAAAAAAAAA", "AAAAAAAAAAGAAAAAAAAAA", "AAAAAAAAAAAAAAAAAAAAA"]),   # IUPAC
        fa(["AAAAAAAAAANAAAAAAAAAA", "AAAAAAAAAAGAAAAAAAAAA", "AAAAAAAAAATAAAAAAAAAA"]),   # N replaced at load
        fa(["ACGTACG-ACGTACG", "ACGTACGTACGTACG", "ACGTACG-ACGTACG", "ACGTACGAACGTACG"]),
        fa(["-----ACGTACGTAAA", "TTTTTACGTACGTAAA", "-----ACGTACGTAAA"]),                 # leading all-gap rows
        fa(["ACGTACGTAAA-----", "ACGTACGTAAATTTTT", "ACGTACGTAAATTTTA"]),
        fa(["AAAAAAAATTAAAAAAAA", "AAAAAAAACCAAAAAAAA", "AAAAAAAAGGAAAAAAAA", "AAAAAAAATTAAAAAAAA", "AAAAAAAATCAAAAAAAA"]),
        fa(["ACGTTGCAACGTTGCAGGCTAGCTAGGATCGATCGATTAGC" * 2, "ACGTTGCATCGTTGCAGGCTAGCTAGGATCCATCGATTAGC" * 2,
            "ACGATGCAACGTTGCAGGCTAGGTAGGATCGATCGATTAGC" * 2, "ACGTTGCAACGTTGGAGGCTAGCTAGGATCGATCGAATAGC" * 2,
            "TCGTTGCAACGTTGCAGGCTACCTAGGATCGATCGATTAGC" * 2, "ACGTTGCAACCTTGCAGGCTAGCTAGGTTCGATCGATTAGC" * 2]),
    ]),
    (5, 3, [
        fa(["AAAATAAAAAA", "AAAACAAAAAA", "CCCCTCCCCCC", "CCCCGCCCCCC"]),
        fa(["AAATAAA", "AAACAAA", "AAA-AAA", "TTTTTTT"]),
    ]),
    (2, 1, [
        fa(["AAACAAAATAAAA", "AAATAAAATAAAA", "AAACAAAAGAAAA", "AAA-------AAA"]),
    ]),
]
