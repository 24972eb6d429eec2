"""Match / non-match interval partition of a consensus string — API of make_prg/from_msa/interval_partition.py,
computed by the device kernel k_partition (mprg_partition)."""
from enum import Enum, auto
from typing import List, Tuple

from ..device import get_backend
from ..engine import BatchEngine, PartitioningError  # noqa: F401
from ..msa import MSA


class IntervalType(Enum):
    Match = auto()
    NonMatch = auto()
    Root = auto()

    @classmethod
    def from_char(cls, letter: str) -> "IntervalType":
        return IntervalType.NonMatch if letter == "*" else IntervalType.Match


def is_type(letter: str, interval_type: IntervalType) -> bool:
    return IntervalType.from_char(letter) is interval_type


class Interval:
    """Closed interval [start, stop] with a type (reference :36-70)."""

    def __init__(self, it_type: IntervalType, start: int, stop: int = None):
        self.type = it_type
        self.start = start
        if stop is not None:
            assert stop >= start
        self.stop = start if stop is None else stop

    def modify_by(self, left_delta: int, right_delta: int):
        self.start += left_delta
        self.stop += right_delta

    def contains(self, position: int) -> bool:
        return self.start <= position <= self.stop

    def __len__(self) -> int:
        return self.stop - self.start + 1

    def __lt__(self, other: "Interval") -> bool:
        return self.start < other.start

    def __eq__(self, other: "Interval") -> bool:
        return self.start == other.start and self.stop == other.stop and self.type is other.type

    def __repr__(self):
        return f"[{self.start}, {self.stop}]"


Intervals = List[Interval]


class IntervalPartitioner:
    """IntervalPartitioner(consensus_string, min_match_length, alignment).get_intervals() (reference :76-126)."""

    def __init__(self, consensus_string: str, min_match_length: int, alignment: MSA):
        self.mml = min_match_length
        triples = BatchEngine(get_backend(), 5, min_match_length).partition(alignment, min_match_length, consensus_string)
        self._match_intervals: Intervals = [Interval(IntervalType.Match, a, b) for a, b, t in triples if t == 0]
        self._non_match_intervals: Intervals = [Interval(IntervalType.NonMatch, a, b) for a, b, t in triples if t == 1]

    def get_intervals(self) -> Tuple[Intervals, Intervals, Intervals]:
        return (sorted(self._match_intervals), sorted(self._non_match_intervals),
                sorted(self._match_intervals + self._non_match_intervals))
