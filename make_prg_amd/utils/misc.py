import os
from itertools import chain, groupby
from typing import Any, List


def remove_duplicated_consecutive_elems_from_list(the_list: List[Any]) -> List[Any]:
    return [key for key, _ in groupby(the_list)]


def flatten_list(list_of_lists: List[List[Any]]) -> List[Any]:
    return list(chain.from_iterable(list_of_lists))


def equal_msas(msa_1, msa_2) -> bool:
    """Alignments are equal when their FASTA renderings are (reference utils/misc.py:16-22)."""
    return format(msa_1, "fasta") == format(msa_2, "fasta")


def should_output_debug_graphs() -> bool:
    return "make_prg_output_debug_graphs" in os.environ
