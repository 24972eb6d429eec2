"""Maximum-likelihood paths of a denovo_paths.txt file (reference make_prg/update/MLPath.py:1-159).

An ML path is the list of PRG leaves a sample's most likely sequence walks through: every node carries the PRG-string
interval of the leaf allele it took (`key`, the key of PrgBuilder.prg_index) and the allele's sequence.  Laid end to end
the node sequences form the sample's linear path; variants are given in that coordinate system.  The nodes tile the
linear path without gaps, so "which node covers position p" is a bisection over the node starts (the reference keeps an
interval tree for the same query)."""
from bisect import bisect_right
from typing import Dict, List, Optional, Tuple


class MLPathError(Exception):
    pass


class EmptyMLPathSequence(Exception):
    pass


class MLPathNode:
    """One `(index [start, end) SEQUENCE)` line (reference MLPath.py:19-68)."""

    def __init__(self, key: Tuple[int, int], sequence: str):
        self.key: Tuple[int, int] = key
        if len(sequence) == 0:
            raise EmptyMLPathSequence(f"Found a ML path node ({self.key}) with empty sequence")
        self.sequence: str = sequence
        self.start_index_in_linear_path: Optional[int] = None      # set by MLPath
        self.end_index_in_linear_path: Optional[int] = None
        if self.key[1] - self.key[0] != len(self.sequence):
            raise MLPathError(f"{self} is not a valid node")

    def __eq__(self, other):
        return isinstance(other, self.__class__) and (self.key, self.sequence) == (other.key, other.sequence)

    def __hash__(self):
        return hash((self.key, self.sequence))

    def __str__(self):
        return (f"PRG key = {self.key}; ML seq interval = [{self.start_index_in_linear_path}:"
                f"{self.end_index_in_linear_path}]; Seq = {self.sequence}")

    def __repr__(self):
        return f'MLPathNode(key={self.key}, sequence="{self.sequence}")'


class MLPath:
    """reference MLPath.py:71-159: indexed by linear-path position and by PRG interval."""

    def __init__(self, ml_path_nodes: List[MLPathNode]):
        if len(ml_path_nodes) == 0:
            raise MLPathError("ML paths cannot be empty")
        self._ml_path_nodes: List[MLPathNode] = ml_path_nodes
        self._starts: List[int] = []
        self._by_prg_interval: Dict[Tuple[int, int], MLPathNode] = {}
        pos = 0
        for node in ml_path_nodes:
            node.start_index_in_linear_path = pos
            pos += len(node.sequence)
            node.end_index_in_linear_path = pos
            self._starts.append(node.start_index_in_linear_path)
            self._by_prg_interval[node.key] = node

    def __eq__(self, other):
        return isinstance(other, self.__class__) and self._ml_path_nodes == other._ml_path_nodes

    def get_last_insertion_pos(self) -> int:
        """Not a covered position, but where a sequence can be inserted after the last base of the last node."""
        return self._ml_path_nodes[-1].end_index_in_linear_path

    def get_last_node(self) -> MLPathNode:
        return self._ml_path_nodes[-1]

    def get_node_given_position_in_linear_path_space(self, position: int) -> MLPathNode:
        i = bisect_right(self._starts, position) - 1
        if i < 0 or position >= self._ml_path_nodes[i].end_index_in_linear_path:
            raise MLPathError(f"No nodes overlap this ML path ({self}) at position {position}, "
                              f"is the denovo_paths.txt given as input correct?")
        return self._ml_path_nodes[i]

    def get_node_given_interval_in_PRG_space(self, interval: Tuple[int, int]) -> MLPathNode:
        if interval not in self._by_prg_interval:
            raise MLPathError(f"PRG space interval ({interval}) not indexed in this node ({self})")
        return self._by_prg_interval[interval]

    def __repr__(self):
        return f"MLPath({self._ml_path_nodes})"
