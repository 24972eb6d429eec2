"""denovo_paths.txt -> per-locus leaf updates (reference make_prg/update/denovo_variants.py:1-559).

File layout (written by pandora discover), per sample and locus: the locus name, the ML path (`N nodes`, then one
`(i [start, end) SEQ)` line per node), `M denovo variants for this locus`, then M tab-separated `pos<TAB>ref<TAB>alt` lines
with 1-based positions in the sample's linear path.  A variant becomes one UpdateData per leaf it touches: the PRG
interval of the leaf allele (key into PrgBuilder.prg_index), the ML path (needed to pad the new sequence with the
sample's alleles in the leaf's other indexed intervals) and the allele with the variant applied."""
import logging
import re
import sys
from collections import Counter, defaultdict, deque
from dataclasses import dataclass
from pathlib import Path
from typing import Deque, Dict, List, Optional, TextIO, Tuple

from ..utils.misc import remove_duplicated_consecutive_elems_from_list
from ..utils.seq_utils import GAP, align
from .ml_path import EmptyMLPathSequence, MLPath, MLPathError, MLPathNode

logger = logging.getLogger("make_prg_amd")


class DenovoError(Exception):
    pass


class NonACGTError(Exception):
    pass


class TooLongDeletion(Exception):
    pass


class DenovoVariant:
    """One `pos ref alt` line, 0-based in the linear path (reference denovo_variants.py:32-314)."""

    def __init__(self, start_index_in_linear_path: int, ref: str, alt: str,
                 ml_path_nodes_it_goes_through: Optional[List[MLPathNode]] = None,
                 long_deletion_threshold: int = sys.maxsize):
        for seq in (ref, alt):
            if any(base not in "ACGT" for base in seq):
                raise NonACGTError(f"Found a non-ACGT seq ({seq}) in a denovo variant")
        if ref == alt:
            raise DenovoError(f"Found a variant where ref ({ref}) equals alt ({alt}), this is not a variant")
        if start_index_in_linear_path < 0:
            raise DenovoError(f"Found a negative index for variant pos ({start_index_in_linear_path})")
        if len(ref) - len(alt) >= long_deletion_threshold:
            raise TooLongDeletion(f"Variant has a too long deletion (delta = {len(ref) - len(alt)}) that should be ignored")
        self.start_index_in_linear_path: int = start_index_in_linear_path
        self.end_index_in_linear_path: int = start_index_in_linear_path + len(ref)
        self.ref: str = ref
        self.alt: str = alt
        self.long_deletion_threshold = long_deletion_threshold
        self.set_ml_path_nodes_it_goes_through(ml_path_nodes_it_goes_through)

    def __eq__(self, other):
        return isinstance(other, self.__class__) and self.__dict__ == other.__dict__

    def is_strict_insertion_event(self) -> bool:
        return len(self.ref) == 0 and len(self.alt) > 0

    def set_ml_path_nodes_it_goes_through(self, nodes: Optional[List[MLPathNode]]):
        """nodes[i] = the ML path node base i of `ref` lies in (a strict insertion names its insertion point only)."""
        if nodes is not None:
            expected = 1 if self.is_strict_insertion_event() else len(self.ref)
            assert len(nodes) == expected, (f"Invalid parameters for DenovoVariant.set_ml_path_nodes_it_goes_through().\n"
                                            f"DenovoVariant: {self}\nml_path_nodes_it_goes_through: {nodes}")
        self.ml_path_nodes_it_goes_through: Optional[List[MLPathNode]] = nodes

    def get_mutated_sequence(self) -> str:
        """The single node this variant lies in, with ref replaced by alt (reference :131-181)."""
        nodes = self.ml_path_nodes_it_goes_through
        assert nodes is not None and len(set(nodes)) == 1, \
            f"Cannot apply variant {self} as it does not go through a single distinct node\nML path nodes: {nodes}"
        node = nodes[0]
        assert (node.start_index_in_linear_path <= self.start_index_in_linear_path
                and self.end_index_in_linear_path <= node.end_index_in_linear_path), \
            f"Node {node} is not compatible with variant {self}"
        lo = self.start_index_in_linear_path - node.start_index_in_linear_path
        hi = lo + len(self.ref)
        assert self.ref == node.sequence[lo:hi], f"Ref is not consistent for {self}. Node = {node}. ref_wrt_indexes = {node.sequence[lo:hi]}"
        return node.sequence[:lo] + self.alt + node.sequence[hi:]

    def _split_variant_at_boundary_alignment(self, ref_alignment: Deque[str], alt_alignment: Deque[str]) -> List["DenovoVariant"]:
        """Walk the ref/alt alignment node by node: the columns that consume a node's bases form its sub-variant
        (reference :183-255)."""
        out = []
        pos = self.start_index_in_linear_path
        bases_in = Counter(self.ml_path_nodes_it_goes_through)
        nodes = remove_duplicated_consecutive_elems_from_list(self.ml_path_nodes_it_goes_through)
        for idx, node in enumerate(nodes):
            sub_ref, sub_alt = [], []
            todo = bases_in[node]
            start = pos
            while todo > 0:
                r = ref_alignment.popleft()
                if r != GAP:
                    sub_ref.append(r)
                    pos += 1
                    todo -= 1
                a = alt_alignment.popleft()
                if a != GAP:
                    sub_alt.append(a)
            if idx == len(nodes) - 1 and len(alt_alignment) > 0:          # trailing inserted bases go to the last node
                sub_alt.extend(a for a in alt_alignment if a != GAP)
            sub_ref_seq, sub_alt_seq = "".join(sub_ref), "".join(sub_alt)
            if sub_ref_seq != sub_alt_seq:
                try:
                    v = DenovoVariant(start, sub_ref_seq, sub_alt_seq, long_deletion_threshold=self.long_deletion_threshold)
                    v.set_ml_path_nodes_it_goes_through([node] * len(sub_ref_seq))
                    out.append(v)
                    logger.debug(f"Split variant to be applied: {v}")
                except TooLongDeletion as error:
                    logger.warning(f"Ignoring split variant: {error}")
        return out

    def split_variant(self) -> List["DenovoVariant"]:
        """Sub-variants that each lie in one node (reference :257-286)."""
        assert self.ml_path_nodes_it_goes_through is not None, \
            "Error on DenovoVariant.split_variant(): self.ml_path_nodes_it_goes_through is None"
        if len(set(self.ml_path_nodes_it_goes_through)) == 1:
            return [self]
        ref_aln, alt_aln = align(self.ref, self.alt)
        return self._split_variant_at_boundary_alignment(deque(ref_aln), deque(alt_aln))

    def __str__(self):
        return f"[{self.start_index_in_linear_path}:{self.end_index_in_linear_path}]:'{self.ref}'->'{self.alt}'"

    def __repr__(self):
        return (f"DenovoVariant(start_index_in_linear_path={self.start_index_in_linear_path}, "
                f'ref="{self.ref}", alt="{self.alt}")')


@dataclass
class UpdateData:
    """What one leaf needs to know about one (sub-)variant (reference :317-326)."""
    ml_path_node_key: Tuple[int, int]
    ml_path: MLPath
    new_node_sequence: str


@dataclass
class DenovoLocusInfo:
    """One locus block of one sample (reference :329-425)."""
    sample: str
    locus: str
    ml_path: MLPath
    variants: List[DenovoVariant]

    def _get_ml_path_nodes_spanning_variant(self, variant: DenovoVariant) -> List[MLPathNode]:
        if variant.is_strict_insertion_event():
            try:
                node = self.ml_path.get_node_given_position_in_linear_path_space(variant.start_index_in_linear_path)
            except MLPathError:
                # an insertion after the last base of the last node
                if variant.start_index_in_linear_path != self.ml_path.get_last_insertion_pos():
                    raise
                node = self.ml_path.get_last_node()
            return [node]
        return [self.ml_path.get_node_given_position_in_linear_path_space(p)
                for p in range(variant.start_index_in_linear_path, variant.end_index_in_linear_path)]

    def get_update_data(self) -> List[UpdateData]:
        out = []
        for variant in self.variants:
            variant.set_ml_path_nodes_it_goes_through(self._get_ml_path_nodes_spanning_variant(variant))
            for sub in variant.split_variant():
                out.append(UpdateData(ml_path_node_key=sub.ml_path_nodes_it_goes_through[0].key, ml_path=self.ml_path,
                                      new_node_sequence=sub.get_mutated_sequence()))
        return out


class DenovoVariantsDB:
    """The whole file: locus name -> list of UpdateData (reference :428-559)."""
    ml_path_regex = re.compile(r"\(\d+ \[(\d+), (\d+)\) ([ACGT]*)\)")

    def __init__(self, filepath: str, long_deletion_threshold: int = sys.maxsize):
        self.filepath: Path = Path(filepath)
        self.long_deletion_threshold = long_deletion_threshold
        with open(self.filepath) as fh:
            by_locus = self._read(fh)
        self.locus_name_to_update_data: Dict[str, List[UpdateData]] = defaultdict(list)
        for locus, infos in by_locus.items():
            for info in infos:
                self.locus_name_to_update_data[locus].extend(info.get_update_data())

    @staticmethod
    def _first_int(fh: TextIO) -> int:
        return int(fh.readline().strip().split()[0])

    @classmethod
    def _read_ml_path(cls, fh: TextIO) -> MLPath:
        nodes = []
        for _ in range(cls._first_int(fh)):
            line = fh.readline().strip()
            m = cls.ml_path_regex.search(line)
            assert m is not None, f"Failed matching ML path regex to line: {line}"
            try:
                nodes.append(MLPathNode(key=(int(m.group(1)), int(m.group(2))), sequence=m.group(3)))
            except EmptyMLPathSequence:
                pass                      # an empty interval cannot be updated
        return MLPath(nodes)

    def _read_variants(self, fh: TextIO) -> List[DenovoVariant]:
        out = []
        for _ in range(self._first_int(fh)):
            cols = fh.readline().strip("\n").split("\t")
            try:
                out.append(DenovoVariant(int(cols[0]) - 1, cols[1], cols[2], long_deletion_threshold=self.long_deletion_threshold))
                logger.debug(f"Read variant: {out[-1]}")
            except (TooLongDeletion, NonACGTError) as error:
                logger.warning(f"Ignoring variant: {error}")
        return out

    def _read(self, fh: TextIO) -> Dict[str, List[DenovoLocusInfo]]:
        by_locus: Dict[str, List[DenovoLocusInfo]] = defaultdict(list)
        try:
            n_samples = self._first_int(fh)
        except IndexError:
            logger.warning(f"File containing denovo paths ({self.filepath}) is empty, is it the correct file?")
            return by_locus
        for _ in range(n_samples):
            sample = fh.readline().strip().split()[1]
            for _ in range(self._first_int(fh)):
                locus = fh.readline().strip()
                ml_path = self._read_ml_path(fh)
                by_locus[locus].append(DenovoLocusInfo(sample, locus, ml_path, self._read_variants(fh)))
        return by_locus
