"""The PRG structure checker used for inputs beyond the oracle's reach (tests/prg_walk.py), checked on the oracle's own PRGs."""
import pytest

import oracle.from_msa_oracle as orc
from make_prg_amd.utils.synthetic import synth_config_fasta, synth_deep_fasta, synth_rows, synth_rows_deep
from tests.prg_walk import check_prg_spells_rows, parse_prg, spellings, _prepare


def test_oracle_prgs_spell_every_input_row_once():
    orc.build_kmeans_lib()
    for rows, N in ((synth_rows(3, 50, 500, 4), 5), (synth_rows_deep(1, 120, 400), 7), (synth_rows(11, 80, 1500, 3), 5)):
        text = "".join(f">s{i}\n{r.decode()}\n" for i, r in enumerate(rows))
        prg, _, _ = orc.build_locus_from_text(text, N, 7)
        assert check_prg_spells_rows(prg, [r.decode() for r in rows]) > 1


def test_checker_rejects_broken_prgs():
    good = "AC 5 G 6 T 5 GG 7 A 8 C 9 T 10 G 9 A 7 T"           # a nested site inside the second allele of site 7
    tree = parse_prg(good)
    _prepare(tree)
    assert spellings(tree, "ACGGGAT") == 1 and spellings(tree, "ACTGGCTAT") == 1 and spellings(tree, "ACTGGCGAT") == 1
    assert spellings(tree, "ACAGGAT") == 0
    dup = "AC 5 G 6 G 5 T"
    t2 = parse_prg(dup)
    _prepare(t2)
    assert spellings(t2, "ACGT") == 2                                # the same allele twice: two paths
    for bad in ("AC 5 G 6 T 7 A 8 C 5 T 7 G", "AC 5 G 6 T", "AC 6 G 5 T", "AC 5 G 5 T", "AC 5 G 6 T 5 A 5 C 6 G 5 "):
        with pytest.raises(AssertionError):
            parse_prg(bad)
