"""Recursion-tree node API of make_prg/recursion_tree.py over the GPU engine.

NodeFactory.build(alignment, prg_builder, parent) runs the whole sub-tree on the device in level-synchronous
passes (make_prg_amd/engine.py) and then materialises the reference's node objects: same classes, attributes,
preorder node ids, nesting levels and traversal output (reference recursion_tree.py:27-572)."""
from abc import ABC, abstractmethod
from typing import List, Optional, Set, Tuple

import numpy as np

from .device import get_backend
from .engine import BatchEngine, NodeRec, decode
from .msa import MSA
from .utils.misc import equal_msas
from .utils.seq_utils import SequenceExpander

SubMSAs = List[MSA]


class UpdateError(Exception):
    pass


class SubAlignment:
    """A node's alignment as a view of its locus' alignment: rows (None = all), column range; all-gap columns are
    dropped when it is turned into an MSA (remove_columns_full_of_gaps_from_MSA, recursion_tree.py:45).  A batch of
    30 000 loci has ~5 million nodes; their alignments are only needed by whoever walks the objects (update, debug,
    pickles), so the nodes keep this descriptor and build the MSA on first use.  All nodes of a locus share `base`,
    so a pickled tree holds the locus' cells once."""
    __slots__ = ("base", "rows", "col0", "ncols")

    def __init__(self, base: MSA, rows, col0: int, ncols: int):
        self.base, self.rows, self.col0, self.ncols = base, rows, col0, ncols

    def materialise(self) -> MSA:
        data = self.base.data
        rows = np.arange(data.shape[0]) if self.rows is None else self.rows
        block = data[rows, self.col0:self.col0 + self.ncols]
        keep = ~(block == ord("-")).all(axis=0)
        return MSA(_data=block[:, keep], _ids=[self.base.ids[r] for r in rows],
                   _descs=[self.base.descriptions[r] for r in rows])


class RecursiveTreeNode(ABC):
    def __init__(self, nesting_level: int, alignment, parent: Optional["RecursiveTreeNode"], prg_builder,
                 children: Optional[List["RecursiveTreeNode"]] = None, node_id: Optional[int] = None):
        self.nesting_level = nesting_level
        self._alignment = alignment           # MSA (already without all-gap columns) or SubAlignment
        self.parent = parent
        self.prg_builder = prg_builder
        self._node_id = prg_builder.get_next_node_id() if node_id is None else node_id
        self._children: List["RecursiveTreeNode"] = [] if children is None else children

    @property
    def alignment(self) -> MSA:
        if isinstance(self._alignment, SubAlignment):
            self._alignment = self._alignment.materialise()
        return self._alignment

    @alignment.setter
    def alignment(self, value):
        self._alignment = value

    @property
    def node_id(self):
        return self._node_id

    @property
    def children(self):
        return self._children

    def __eq__(self, other) -> bool:
        if (self.nesting_level, self.prg_builder.locus_name, self.node_id) != \
                (other.nesting_level, other.prg_builder.locus_name, other.node_id):
            return False
        if (self.parent is None) != (other.parent is None):
            return False
        if self.parent is not None and self.parent.node_id != other.parent.node_id:
            return False
        if not equal_msas(self.alignment, other.alignment):
            return False
        if len(self.children) != len(other.children):
            return False
        return all(a == b for a, b in zip(self.children, other.children))

    def __hash__(self):
        return hash((self.node_id, self.prg_builder.locus_name))

    @abstractmethod
    def preorder_traversal_to_build_prg(self, prg_as_list: List[str], delim_char: str = " "):
        raise NotImplementedError

    def is_leaf(self) -> bool:
        return len(self.children) == 0

    def is_root(self) -> bool:
        return self.parent is None

    def replace_child(self, old_child, new_child):
        assert old_child in self.children, f"Failure to replace a child, {old_child} does not exist"
        self.children[self.children.index(old_child)] = new_child

    def __repr__(self):
        return (f"{self.__class__.__name__}:\nId = {self.node_id}\nNesting level = {self.nesting_level}\n"
                f"Parent = {'None' if self.parent is None else f'Id = {self.parent.node_id}'}\n"
                f"Children = [{', '.join(f'Id = {child.node_id}' for child in self.children)}]\n"
                f"Alignment:\n{format(self.alignment, 'fasta')}")

    __str__ = __repr__


class MultiIntervalNode(RecursiveTreeNode):
    """Vertical partition: PRG = concatenation of the children's PRGs (reference :176-201)."""

    def preorder_traversal_to_build_prg(self, prg_as_list: List[str], delim_char: str = " "):
        for child in self.children:
            child.preorder_traversal_to_build_prg(prg_as_list, delim_char)


class MultiClusterNode(RecursiveTreeNode):
    """Horizontal partition: opens a site, one allele per child (reference :204-239)."""

    def preorder_traversal_to_build_prg(self, prg_as_list: List[str], delim_char: str = " "):
        site = self.prg_builder.get_next_site_num()
        prg_as_list.extend(f"{delim_char}{site}{delim_char}")
        last = len(self.children) - 1
        for i, child in enumerate(self.children):
            child.preorder_traversal_to_build_prg(prg_as_list, delim_char)
            prg_as_list.extend(f"{delim_char}{site + 1 if i < last else site}{delim_char}")


class LeafNode(RecursiveTreeNode):
    """Never partitioned; the only nodes that get indexed and updated (reference :246-391)."""

    def __init__(self, nesting_level, alignment, parent, prg_builder, node_id=None):
        super().__init__(nesting_level, alignment, parent, prg_builder, [], node_id)
        self.new_sequences: Set[str] = set()
        self.indexed_PRG_intervals: Set[Tuple[int, int]] = set()

    def preorder_traversal_to_build_prg(self, prg_as_list: List[str], delim_char: str = " ", do_indexing=True):
        seqs = SequenceExpander.get_expanded_sequences_from_MSA(self.alignment)
        if len(seqs) == 1:
            start = len(prg_as_list)
            prg_as_list.extend(seqs[0])
            if do_indexing:
                self.prg_builder.update_PRG_index(start, len(prg_as_list), node=self)
            return
        site = self.prg_builder.get_next_site_num()
        prg_as_list.extend(f"{delim_char}{site}{delim_char}")
        for i, seq in enumerate(seqs):
            start = len(prg_as_list)
            prg_as_list.extend(seq)
            end = len(prg_as_list)
            prg_as_list.extend(f"{delim_char}{site + 1 if i < len(seqs) - 1 else site}{delim_char}")
            if do_indexing:
                self.prg_builder.update_PRG_index(start, end, node=self)

    # ---- update hooks (reference :302-391)
    def add_data_to_batch_update(self, update_data):
        """One (sub-)variant that falls into this leaf: the new allele, padded with the sample's alleles in the leaf's
        other indexed PRG intervals (a leaf with several alleles is indexed once per allele) — reference :304-342."""
        from .update.ml_path import MLPathError
        key = update_data.ml_path_node_key
        if key not in self.indexed_PRG_intervals:
            raise UpdateError(f"PRG interval {key} not found in indexed PRG intervals for node: {self.indexed_PRG_intervals}")
        parts = []
        for interval in sorted(self.indexed_PRG_intervals):
            if interval == key:
                parts.append(update_data.new_node_sequence)
            else:
                try:
                    parts.append(update_data.ml_path.get_node_given_interval_in_PRG_space(interval).sequence)
                except MLPathError:
                    pass
        self.new_sequences.add("".join(parts))

    def add_indexed_PRG_interval(self, interval: Tuple[int, int]):
        self.indexed_PRG_intervals.add(interval)

    def clear_PRG_interval_index(self):
        self.indexed_PRG_intervals.clear()

    def batch_update(self):
        if self.new_sequences:
            self._update_leaf()

    def updated_alignment(self) -> MSA:
        """The aligner's part of _update_leaf (reference :366-371)."""
        assert self.prg_builder.aligner is not None, "Cannot make updates without a Multiple Sequence Aligner."
        return self.prg_builder.aligner.get_updated_alignment(current_alignment=self.alignment,
                                                              new_sequences=self.new_sequences)

    def replace_by(self, new_node: "RecursiveTreeNode"):
        """The tree surgery of _update_leaf (reference :378-388): the rebuilt sub-tree takes this leaf's place and the
        builder's PRG index, now stale to the right of the site, is cleared."""
        if self.is_root():
            self.prg_builder.replace_root(new_node)
        else:
            self.parent.replace_child(self, new_node)
        self.prg_builder.clear_PRG_index()

    def _update_leaf(self):
        self.replace_by(NodeFactory.build(self.updated_alignment(), self.prg_builder, self.parent))


class NodeFactory:
    @staticmethod
    def build(alignment: MSA, prg_builder, parent_node: Optional[RecursiveTreeNode] = None) -> RecursiveTreeNode:
        """reference :401-471 — leaf / multi-interval / multi-cluster decision and the whole sub-tree below it."""
        eng = BatchEngine(get_backend(), prg_builder.max_nesting, prg_builder.min_match_length)
        eng.load([alignment])
        res = eng.run(root_level=0 if parent_node is None else parent_node.nesting_level,
                      root_is_tree_root=parent_node is None)[0]
        if res.error is not None:
            raise res.error
        return materialise(eng, res, alignment, prg_builder, parent_node)

    # ---- the reference's private helpers, same names and results (reference :475-572); the batched build does not go
    #      through them (one level of many alignments per launch), callers and the reference's unit tests do
    @staticmethod
    def _get_vertical_partition(alignment: MSA, min_match_length: int):
        """(all intervals, match intervals) of the alignment's consensus — reference :500-513."""
        from .from_msa.interval_partition import IntervalPartitioner
        from .utils.seq_utils import get_consensus_from_MSA
        match, _non_match, all_intervals = IntervalPartitioner(get_consensus_from_MSA(alignment), min_match_length,
                                                               alignment).get_intervals()
        return all_intervals, match

    @staticmethod
    def _is_multi_interval(all_intervals) -> bool:
        return len(all_intervals) > 1

    @staticmethod
    def _is_single_match_interval(all_intervals, match_intervals) -> bool:
        return len(all_intervals) == 1 and all_intervals[0] in match_intervals

    @staticmethod
    def _partition_alignment_into_interval_subalignments(alignment: MSA, all_intervals) -> SubMSAs:
        return [alignment[:, interval.start:interval.stop + 1] for interval in all_intervals]

    @staticmethod
    def _alignment_has_issues(alignment: MSA) -> bool:
        """Too few distinct sequences to cluster, or one sequence aligned in two ways — reference :475-494."""
        from .utils.seq_utils import (get_number_of_unique_gapped_sequences, get_number_of_unique_ungapped_sequences)
        n_ungapped = get_number_of_unique_ungapped_sequences(alignment)
        return n_ungapped <= 2 or n_ungapped < get_number_of_unique_gapped_sequences(alignment)

    @staticmethod
    def _infer_if_we_should_cluster_further(alignment: MSA, clustering_result, nesting_level: int, max_nesting: int) -> bool:
        """reference :538-556."""
        if clustering_result.no_clustering or nesting_level + 1 >= max_nesting:
            return False
        return not NodeFactory._alignment_has_issues(alignment)

    @staticmethod
    def _get_sub_alignment_by_list_id(id_list, alignment: MSA) -> MSA:
        wanted = set(id_list)
        return MSA([record for record in alignment if record.id in wanted])

    @staticmethod
    def _get_subalignments_by_clustering(alignment: MSA, clustering_result) -> SubMSAs:
        """One sub-alignment per cluster, rows in the alignment's order — reference :558-572."""
        return [NodeFactory._get_sub_alignment_by_list_id(ids, alignment) for ids in clustering_result.clustered_ids]

    @staticmethod
    def build_many(jobs) -> list:
        """Batched re-entry: jobs = [(alignment, prg_builder, parent_node or None), ...] -> the sub-trees, in job order.
        What the reference's `update` does leaf by leaf (LeafNode._update_leaf -> NodeFactory.build,
        recursion_tree.py:374-376) for every touched leaf of every locus is here ONE resident batch per
        (max_nesting, min_match_length): every alignment starts at its parent's nesting level, none but a tree root
        is forced to be a MultiIntervalNode.  Node ids are drawn from each job's builder in job order, as a
        sequential run would.  A job whose alignment fails (SequenceCurationError ...) yields the exception object."""
        out = [None] * len(jobs)
        groups = {}
        for i, (aln, builder, parent) in enumerate(jobs):
            groups.setdefault((builder.max_nesting, builder.min_match_length), []).append(i)
        for (max_nesting, min_match_length), idxs in groups.items():
            eng = BatchEngine(get_backend(), max_nesting, min_match_length)
            eng.load([jobs[i][0] for i in idxs])
            results = eng.run(root_level=[0 if jobs[i][2] is None else jobs[i][2].nesting_level for i in idxs],
                              root_is_tree_root=[jobs[i][2] is None for i in idxs])
            for i, res in zip(idxs, results):
                out[i] = (eng, res)
        for i, (aln, builder, parent) in enumerate(jobs):          # materialise in job order: ids come from the builders
            eng, res = out[i]
            out[i] = res.error if res.error is not None else materialise(eng, res, aln, builder, parent)
        return out


def materialise(eng: BatchEngine, res, alignment: MSA, prg_builder, parent_node=None) -> RecursiveTreeNode:
    """Engine records → reference-style node objects; ids drawn from the builder in preorder (reference :48-55)."""
    nodes = res.nodes
    codes_ascii = alignment.data

    def make(ni: int, parent) -> RecursiveTreeNode:
        nd: NodeRec = nodes[ni]
        rows = np.arange(codes_ascii.shape[0]) if nd.rows is None else nd.rows
        block = codes_ascii[rows, nd.col0:nd.col0 + nd.ncols][:, nd.keep_cols]
        stored = MSA(_data=block, _ids=[alignment.ids[r] for r in rows],
                     _descs=[alignment.descriptions[r] for r in rows])
        if nd.kind == "leaf":
            return LeafNode(nd.level, stored, parent, prg_builder)
        cls = MultiIntervalNode if nd.kind == "interval" else MultiClusterNode
        node = cls(nd.level, stored, parent, prg_builder, [])
        for c in nd.children:
            node._children.append(make(c, node))
        assert not node.is_leaf(), f"{cls.__name__}s should never be leaves"
        return node

    return make(res.root, parent_node)


def materialise_forest(eng, mi: int, alignment: MSA, prg_builder, leaf_of: Optional[dict] = None) -> RecursiveTreeNode:
    """Same as materialise() for one tree of a forest.ForestEngine batch (assemble_prgs() must have run).
    leaf_of (optional dict) receives {node id: LeafNode} so that the caller can attach the batch's PRG index."""
    from .forest import KIND_INTERVAL, KIND_LEAF
    t = eng.tab
    node_id = eng.node_id
    data = alignment.data
    pool = eng.pool_host() if eng.pool_used else np.zeros(0, np.int64)

    def make(ni: int, parent) -> RecursiveTreeNode:
        rows = eng.node_rows(ni, pool)
        stored = SubAlignment(alignment, None if rows is None else rows.copy(), int(t["col0"][ni]), int(t["ncols"][ni]))
        level, kind = int(t["level"][ni]), int(t["kind"][ni])
        if kind == KIND_LEAF:
            leaf = LeafNode(level, stored, parent, prg_builder)
            if leaf_of is not None:
                leaf_of[int(node_id[ni])] = leaf
            return leaf
        node = (MultiIntervalNode if kind == KIND_INTERVAL else MultiClusterNode)(level, stored, parent, prg_builder, [])
        for j in range(int(t["n_child"][ni])):
            node._children.append(make(int(t["first_child"][ni]) + j, node))
        return node

    return make(int(eng.root_of[mi]), None)
