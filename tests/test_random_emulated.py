"""Seeded random small alignments: both hosts (array-at-a-time forest, node objects) on the emulation backend
must agree with the oracle on PRG, recursion tree and prg_index, for several (max_nesting, min_match_length)."""
import pytest

from tests import parity_common as pc
from tests.emu.backend import EmuBackend
from tests.random_msas import random_cases


@pytest.fixture(scope="module")
def emu():
    return EmuBackend()


@pytest.mark.parametrize("N,L,seed", [(5, 7, 11), (5, 3, 12), (2, 1, 13), (1, 7, 14), (5, 2, 15)])
def test_forest_host(emu, N, L, seed, monkeypatch):
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(emu, random_cases(seed, 150), N, L)


@pytest.mark.parametrize("N,L,seed", [(5, 7, 21), (3, 3, 22)])
def test_node_host(emu, N, L, seed, monkeypatch):
    monkeypatch.setattr(pc, "ENGINE", "nodes")
    pc.check_vs_oracle(emu, random_cases(seed, 80), N, L)


def _wide_fasta(seed, S, C, p_mut, gaps=True):
    import numpy as np
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, C)
    out = []
    for i in range(S):
        y = base.copy()
        m = rng.random(C) < p_mut
        y[m] = rng.integers(0, 4, int(m.sum()))
        txt = np.frombuffer(b"ACGT", np.uint8)[y].copy()
        if gaps:
            for st in np.nonzero(rng.random(C) < 0.003)[0]:
                txt[st:st + int(rng.integers(1, 5))] = ord("-")
        out.append(f">w{i}\n{txt.tobytes().decode()}\n")
    return "".join(out)


@pytest.mark.parametrize("N,L,S,C,p", [(2, 1, 5, 900, 0.02),       # n/(L-1) exceeds the LDS interval stacks: global stacks
                                       (2, 2, 4, 1400, 0.05),
                                       (1, 7, 3, 13000, 0.002)])   # wider than the LDS column bytes (PT_COLS): masks read directly
def test_wide_views_take_the_fallback_paths(emu, N, L, S, C, p, monkeypatch):
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(emu, [_wide_fasta(100 + C, S, C, p)], N, L)


def test_gap_runs_reach_across_column_segments(emu, monkeypatch):
    """k_gap_runs splits a wide view into 512-column segments: a run that begins before a segment (or covers whole segments) is
    counted backwards from the segment's start.  Rows with gap stretches across columns 2 048 and 4 096, variation inside them."""
    import numpy as np
    rng = np.random.default_rng(5)
    C = 4400
    base = rng.integers(0, 4, C)
    rows = []
    for i in range(6):
        y = np.frombuffer(b"ACGT", np.uint8)[base].copy()
        for c in (1990, 2040, 2100, 3000, 4090, 4100, 4300):          # variation: non-match columns inside and next to the stretches
            y[c] = b"ACGT"[(int(base[c]) + 1 + i % 3) % 4]
        rows.append(y)
    rows[1][2030:2060] = ord("-")          # across the first boundary
    rows[2][1985:4200] = ord("-")          # a whole segment and both boundaries
    rows[4][4096:4110] = ord("-")          # begins exactly at a boundary
    rows[5][4080:4096] = ord("-")          # ends exactly before one
    text = "".join(f">g{i}\n{r.tobytes().decode()}\n" for i, r in enumerate(rows))
    monkeypatch.setattr(pc, "ENGINE", "forest")
    for N, L in ((3, 7), (2, 3)):
        pc.check_vs_oracle(emu, [text], N, L)


def test_big_view_row_groups_when_every_hash_collides(monkeypatch):
    """More rows than k_ungap_dedupe's LDS table holds: k_dedupe_scan_big decides by symbols behind the hash filter — here with the
    test-only hash that only counts symbols, so most candidates pass the filter and differ."""
    import numpy as np
    weak = EmuBackend(defines=("MPRG_TEST_WEAK_HASH",), tag="_weakhash")
    rng = np.random.default_rng(9)
    C = 40
    variants = [rng.integers(0, 4, C) for _ in range(9)]
    rows = [variants[int(rng.integers(0, len(variants)))].copy() for _ in range(700)]
    txt = [np.frombuffer(b"ACGT", np.uint8)[r].copy() for r in rows]
    for i in range(0, 700, 7):
        txt[i][5:8] = ord("-")          # gapped twins of ungapped-different rows, and the reverse
    text = "".join(f">b{i}\n{t.tobytes().decode()}\n" for i, t in enumerate(txt))
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(weak, [text], 4, 5)


def test_leaf_of_many_alleles_is_laid_out_by_its_wavefront(emu, monkeypatch):
    """k_as_leaf_jobs: a leaf of more than 128 alleles (here the child of a root at the nesting limit, 300 and 90 distinct rows in one
    batch so that big and small leaves share wavefronts) gets its alleles' places from a wavefront prefix sum."""
    import numpy as np
    rng = np.random.default_rng(21)
    texts = []
    for S in (300, 90, 200):
        C = 30
        base = rng.integers(0, 4, C)
        rows = []
        for i in range(S):
            y = base.copy()
            y[5:25] = rng.integers(0, 4, 20)
            t = np.frombuffer(b"ACGT", np.uint8)[y].copy()
            if i % 5 == 0:
                t[10:10 + i % 7] = ord("-")          # alleles of different lengths
            rows.append(t.tobytes().decode())
        texts.append("".join(f">a{i}\n{r}\n" for i, r in enumerate(rows)))
    monkeypatch.setattr(pc, "ENGINE", "forest")
    eng = pc.check_vs_oracle(emu, texts, 1, 7)
    assert int(eng.tab["nseq"].max()) > 128


def test_tall_view_takes_the_wide_majority_workgroups(emu, monkeypatch):
    """More rows than the LDS member lists of k_cluster_majority hold (CF_ROWS): k_cluster_majority_big's shared counters."""
    import numpy as np
    rng = np.random.default_rng(77)
    C = 48
    clades = [rng.integers(0, 4, C) for _ in range(3)]
    variants = []
    for cl in clades:
        for _ in range(4):
            y = cl.copy()
            y[rng.integers(0, C, 2)] = rng.integers(0, 4, 2)
            variants.append(y)
    rows = [variants[int(rng.integers(0, len(variants)))] for _ in range(1100)]
    text = "".join(f">t{i}\n{np.frombuffer(b'ACGT', np.uint8)[r].tobytes().decode()}\n" for i, r in enumerate(rows))
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(emu, [text], 5, 7)


def test_row_grouping_is_exact_when_every_hash_collides(monkeypatch):
    """Test-only build whose row hash only counts symbols: the nominee check refutes, the exact search decides."""
    weak = EmuBackend(defines=("MPRG_TEST_WEAK_HASH",), tag="_weakhash")
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(weak, random_cases(31, 120), 5, 7)
    pc.check_vs_oracle(weak, random_cases(32, 60), 3, 2)
    # the same build lets every k-mer hash collide under the first two seeds (k-mer sizes > 16): the dictionary is rebuilt until
    # a seed separates the k-mers, and the answers stay the real reference's
    from tests.long_kmer_common import check_long_kmers
    assert check_long_kmers(weak) == 4


def test_more_clusters_than_the_lds_offsets_of_split_children(emu, monkeypatch):
    """1 100 distinct sequences shorter than the k-mer size next to 40 long ones: every short one is a cluster of its own,
    more ranks than k_split_children keeps offsets for in LDS (SC_RANKS) — the one-lane form."""
    import itertools
    import numpy as np
    monkeypatch.setattr(pc, "ENGINE", "forest")
    rng = np.random.default_rng(3)
    rows = [f"ACGTTGCAAC{''.join(w)}------GGATCCATGA" for w in itertools.islice(itertools.product("ACGT", repeat=6), 1100)]
    for i in range(40):
        base = list(("ACGTACGTACGT", "TTGACCTGAATC")[i % 2])
        base[int(rng.integers(0, 12))] = "ACGT"[int(rng.integers(0, 4))]
        rows.append("ACGTTGCAAC" + "".join(base) + "GGATCCATGA")
    text = "".join(f">s{i}\n{r}\n" for i, r in enumerate(rows))
    eng = pc.check_vs_oracle(emu, [text], 5, 7)
    assert np.bincount(eng.tab["parent"][eng.tab["parent"] >= 0]).max() > 1024      # a cluster node with > 1024 children


def test_clustering_loop_forms_agree(monkeypatch, golden_integration):
    """The fused loop's cluster_further has an in-LDS form (the view's cells fit the fit's pool) and a global-memory form with row
    slices; the per-round launches of rounds 1-3 are a third way through the same decisions.  A test-only build forces the
    global form; MPRG_KLOOP=rounds the per-round host: same answers as the oracle / the real reference's goldens."""
    import make_prg_amd.forest as F
    glob = EmuBackend(defines=("MPRG_TEST_CF_GLOBAL",), tag="_cfglobal")
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(glob, random_cases(41, 60), 5, 7)
    assert pc.check_integration(glob, golden_integration) >= 30
    monkeypatch.setattr(F, "KLOOP", "rounds")
    pc.check_vs_oracle(EmuBackend(), random_cases(42, 40), 5, 7)
    assert pc.check_integration(EmuBackend(), golden_integration) >= 30


def test_problems_prepared_without_tables_stay_with_the_wide_fits(emu, monkeypatch):
    """A level of big problems prepares the matrices beyond the prepare kernels' LDS (156 KB) by mprg_kmeans_prepare_big — WITHOUT the
    seeding's sample-sample tables from forest.KM_NO_TABLES_BYTES on.  The LDS form of the fits reads those tables: the forest's control steps (kml_class_ws)
    leave such problems to the wide fits even where their restarts' state would fit its largest classes (here: 44 distinct sequences whose
    ~750 k-mers make a 260 KB matrix; five close clades + noise, so that a seeding from garbage tables changes the answer — a build without
    the rule fails this test).  Thresholds of 1 byte: every level counts as big, no big problem gets tables."""
    import numpy as np
    import make_prg_amd.forest as F
    monkeypatch.setattr(F, "KM_BIG_BYTES", 1)
    monkeypatch.setattr(F, "KM_NO_TABLES_BYTES", 1)
    monkeypatch.setattr(pc, "ENGINE", "forest")
    texts = []
    for seed in (92, 97):
        rng = np.random.default_rng(seed)
        C = 44
        base = rng.integers(0, 4, C)
        clades = []
        for _ in range(5):
            y = base.copy(); m = rng.random(C) < 0.3; y[m] = rng.integers(0, 4, int(m.sum())); clades.append(y)
        rows = []
        for i in range(44):
            y = clades[i % 5].copy(); m = rng.random(C) < 0.1; y[m] = rng.integers(0, 4, int(m.sum()))
            rows.append(np.frombuffer(b"ACGT", np.uint8)[y].tobytes().decode())
        texts.append("".join(f">w{i}\n{r}\n" for i, r in enumerate(rows)))
    eng = pc.check_vs_oracle(emu, texts, 2, 7)
    assert eng._big_seen and int(eng.counters.get("max_problem_bytes", 0)) > 156 * 1024
    assert emu.lib.mprg_kmeans_lds_class(44, 744, 2, 10) >= 0          # (by its shape the top problem HAS a class: the workspace's flag decides)
    # ... and with the tables made (the default threshold) the same problems take the LDS form: same answers
    monkeypatch.setattr(F, "KM_NO_TABLES_BYTES", 4 << 30)
    pc.check_vs_oracle(emu, texts, 2, 7)


def test_kmeans_forms_of_earlier_rounds_through_the_forest(monkeypatch, golden_integration):
    """The default since round 6 is the LDS form of the fits (KM_MODE bit 2: k_kmeans_fit_lds per round, k_cluster_loop_lds fused).  The forms
    it replaced stay entry points of the ABI: the small / general workgroup forms (KM_MODE = 2) and the general form alone (0), fused and
    per round — same answers as the oracle / the real reference's goldens."""
    import make_prg_amd.forest as F
    monkeypatch.setattr(pc, "ENGINE", "forest")
    for mode, loop in ((2, "fused"), (2, "rounds"), (0, "fused")):
        monkeypatch.setattr(F, "KM_MODE", mode)
        monkeypatch.setattr(F, "KM_LDS_ENTRY", "mprg_kmeans_fit_wave")
        monkeypatch.setattr(F, "KM_LISTS", ((F.KM_LDS_ENTRY, 0), (F.KM_LDS_ENTRY, 1), (F.KM_LDS_ENTRY, 2), (F.KM_LDS_ENTRY, 3), ("mprg_kmeans_fit", None),
                                            ("mprg_kmeans_fit_small", 0), ("mprg_kmeans_fit_small", 1)))
        monkeypatch.setattr(F, "KLOOP", loop)
        pc.check_vs_oracle(EmuBackend(), random_cases(43 + mode, 24), 5, 7)
        assert pc.check_integration(EmuBackend(), golden_integration) >= 30


def test_levels_with_big_problems_take_the_wide_fits(emu, monkeypatch, golden_integration):
    """forest.KM_BIG_BYTES: a level whose largest count matrix reaches it runs the per-round loop with a wide workgroup per restart
    for its general-form fits (mprg_kmeans_fit_wide) — here every level (threshold 1 byte), in an engine whose loop is otherwise the
    fused one; no plan is kept for such a batch.  Same answers as the oracle / the real reference's goldens."""
    import make_prg_amd.forest as F
    monkeypatch.setattr(F, "KM_BIG_BYTES", 1)
    monkeypatch.setattr(F, "KM_MODE", 0)          # (every fit through the general form)
    monkeypatch.setattr(pc, "ENGINE", "forest")
    eng = pc.check_vs_oracle(emu, random_cases(51, 24), 5, 7)
    assert eng._big_seen and eng._plan is None
    assert pc.check_integration(emu, golden_integration) >= 30
    # the same with the wide workgroups reading the counts as bytes from the workspace (test-only build: no fit "fits the LDS pool")
    glob = EmuBackend(defines=("MPRG_TEST_WIDE_GLOBAL",), tag="_wideglobal")
    pc.check_vs_oracle(glob, random_cases(52, 24), 5, 7)
    assert pc.check_integration(glob, golden_integration) >= 30


@pytest.mark.parametrize("km_mode", [0, 2])
def test_every_round_of_a_big_level_at_once(emu, monkeypatch, golden_integration, km_mode):
    """forest.KM_SPEC_PROBLEMS: a big level of few problems launches the general-form fit of EVERY k = 2..10 at once (wide workgroups,
    restart slots and result slices per k) and the loop's rounds settle on results that are already there — here every level
    (threshold 1 byte, any number of problems), with every fit in the general form (km_mode 0) and with the small fits in their own
    per-round launches beside the speculative ones (km_mode 2).  Same answers as the oracle / the real reference's goldens."""
    import make_prg_amd.forest as F
    monkeypatch.setattr(F, "KM_BIG_BYTES", 1)
    monkeypatch.setattr(F, "KM_SPEC_PROBLEMS", 1 << 30)
    monkeypatch.setattr(F, "KM_MODE", km_mode)
    monkeypatch.setattr(pc, "ENGINE", "forest")
    eng = pc.check_vs_oracle(emu, random_cases(53, 24), 5, 7)
    assert eng._big_seen and eng.counters.get("speculative_levels", 0) >= 1
    assert pc.check_integration(emu, golden_integration) >= 30


@pytest.mark.parametrize("no_tables_from", [1, 1 << 40])
def test_big_problem_through_the_byte_matrix(emu, monkeypatch, no_tables_from):
    """A clustering problem whose count matrix is beyond the LDS prepare classes (150 distinct sequences x ~250 4-mers: 300 KB) in a level
    that counts as big: mprg_kmeans_prepare_big writes the byte matrix, the wide fits stream it — with the seeding's tables
    (no_tables_from = huge) and without (1: their elements on demand)."""
    import numpy as np
    import make_prg_amd.forest as F
    monkeypatch.setattr(F, "KM_BIG_BYTES", 200_000)
    monkeypatch.setattr(F, "KM_NO_TABLES_BYTES", no_tables_from)
    monkeypatch.setattr(pc, "ENGINE", "forest")
    rng = np.random.default_rng(17)
    clades = [rng.integers(0, 4, 34) for _ in range(3)]
    rows = []
    for i in range(150):
        y = clades[i % 3].copy()
        m = rng.random(34) < 0.25
        y[m] = rng.integers(0, 4, int(m.sum()))
        rows.append("ACGTACGT" + np.frombuffer(b"ACGT", np.uint8)[y].tobytes().decode() + "TTGACCAT")
    text = "".join(f">q{i}\n{r}\n" for i, r in enumerate(rows))
    eng = pc.check_vs_oracle(emu, [text], 2, 4)
    assert eng._big_seen and eng.counters["max_problem_bytes"] > 156 * 1024


def test_kmer_dictionary_by_many_workgroups(emu, monkeypatch):
    """mprg_kmer_dictionary_parts (the form of problems with millions of k-mer occurrences) for every level: same ids — the column
    order of the count matrices, hence every KMeans sum — as the one-workgroup dictionary; oracle answers."""
    import make_prg_amd.forest as F
    monkeypatch.setattr(F, "KD_PARTS_FROM", 1)
    monkeypatch.setattr(F, "KD_PARTS", 5)
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(emu, random_cases(61, 60), 5, 7)
    pc.check_vs_oracle(emu, random_cases(62, 30), 3, 3)


def test_wide_and_tall_view_shares_a_rows_candidates_among_threads(emu, monkeypatch):
    """More than 4 096 columns AND more rows than k_ungap_dedupe's LDS table: k_dedupe_scan_big's chunks are 8 rows and 32 threads
    share a row's earlier rows (candidate classes); the view's rows are wide enough for k_cluster_hamming's group-per-row form and
    k_ungap_hash's batched steps.  Four variants (one mutated column in every five, at different phases: the whole alignment is ONE
    non-match interval), 530 rows drawn from them, a few with gaps (gapped twins of rows that are equal without gaps)."""
    import numpy as np
    rng = np.random.default_rng(31)
    C, S = 4200, 530
    base = rng.integers(0, 4, C)
    variants = [base.copy()]
    for v in range(3):
        y = base.copy()
        cols = np.arange(C)[np.arange(C) % 5 == v + 1]
        y[cols] = (y[cols] + 1 + v) % 4
        variants.append(y)
    pick = rng.integers(0, len(variants), S)
    pick[:8] = [3, 1, 3, 0, 2, 1, 0, 2]          # every variant's first row early, repeats in different candidate classes
    txt = [np.frombuffer(b"ACGT", np.uint8)[variants[int(p)]].copy() for p in pick]
    for i in range(5, S, 37):
        txt[i][100 + i % 50:103 + i % 50] = ord("-")
    text = "".join(f">w{i}\n{t.tobytes().decode()}\n" for i, t in enumerate(txt))
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(emu, [text], 2, 7)


def test_sample_tables_by_tiles_equal_the_chains_by_threads(emu):
    """k_kmeans_prepare_tables_tiled advances all chains of a sample pair together over LDS-staged features; the tables must be the
    doubles of k_kmeans_prepare_tables (a thread per element walking km_gemm_dot_v / km_chain4 / km_gemv_col / km_euclid): shapes on
    both sides of every block boundary (256 / 512 of the dgemm K loop, 2 048 of dgemv_t, V mod 4 and mod 8, D mod 4 and the 32 x 32 tiles)."""
    from tests.kmeans_tables import check_lds_tables, check_tiled_tables
    assert check_tiled_tables(emu) == []
    assert check_lds_tables(emu) == []          # K6's LDS form: a thread per pair over the matrix in LDS
