"""Seeded generator of small, nasty alignments: gaps (runs, all-gap columns, all-gap rows in a stretch), ambiguity
codes, N, lower case, duplicated rows, tiny widths — the situations the reference's unit tests probe one by one."""
import numpy as np


def random_fasta(rng) -> str:
    S = int(rng.integers(1, 14))
    C = int(rng.integers(1, 60))
    style = int(rng.integers(0, 4))
    base = rng.integers(0, 4, C)
    n_clades = int(rng.integers(1, 4))
    clades = []
    for _ in range(n_clades):
        y = base.copy()
        m = rng.random(C) < (0.02, 0.1, 0.3, 0.6)[style]
        y[m] = rng.integers(0, 4, int(m.sum()))
        clades.append(y)
    rows = []
    for _ in range(S):
        y = clades[int(rng.integers(0, n_clades))].copy()
        m = rng.random(C) < (0.01, 0.05, 0.1, 0.02)[style]
        y[m] = rng.integers(0, 4, int(m.sum()))
        txt = np.frombuffer(b"ACGT", np.uint8)[y].copy()
        for st in np.nonzero(rng.random(C) < 0.04)[0]:
            txt[st:st + int(rng.integers(1, 8))] = ord("-")
        if rng.random() < 0.15:
            for st in np.nonzero(rng.random(C) < 0.03)[0]:
                txt[st] = ord("RYKMSW"[int(rng.integers(0, 6))])
        if rng.random() < 0.15:
            for st in np.nonzero(rng.random(C) < 0.03)[0]:
                txt[st] = ord("N")
        rows.append(txt)
    if S > 2 and rng.random() < 0.3:
        rows[int(rng.integers(1, S))] = rows[0].copy()                    # duplicate row
    if rng.random() < 0.2 and C > 3:
        c = int(rng.integers(0, C))
        for r in rows:
            r[c] = ord("-")                                                # all-gap column
    if rng.random() < 0.1:
        rows[int(rng.integers(0, S))][:] = ord("-")                        # an empty row
    if rng.random() < 0.04:
        rows[int(rng.integers(0, S))][int(rng.integers(0, C))] = ord("X")  # disallowed base: the locus is skipped
    out = []
    for i, r in enumerate(rows):
        s = r.tobytes().decode()
        if rng.random() < 0.1:
            s = s.lower()
        out.append(f">s{i} sample {i}\n{s}\n")
    return "".join(out)


def random_cases(seed: int, n: int):
    rng = np.random.default_rng(seed)
    return [random_fasta(rng) for _ in range(n)]
