"""Sequence clustering of a sub-alignment — API of make_prg/from_msa/cluster_sequences.py:1-296.

kmeans_cluster_seqs() is the function the recursion calls: k-mer featurisation, KMeans and the one-reference-like test
run on the device (mprg_kmer_*, mprg_kmeans_*, mprg_cluster_further) with the id bookkeeping on the host
(engine.BatchEngine.cluster).  The small helpers the reference module also exports — callers and the reference's unit
tests import them — are kept with the same names, arguments and errors; sequences_are_one_reference_like() and
cluster_further() go through the same device kernel as the recursion when a backend is active, the rest are a few lines
of host code."""
from collections import Counter
from dataclasses import dataclass
from itertools import chain
from typing import Dict, Iterator, List, Optional, Union

import numpy as np

from ..device import get_backend
from ..engine import BatchEngine, expand_sequences
from ..msa import MSA

DISTANCE_THRESHOLD: float = 0.2
LENGTH_THRESHOLD: int = 5
MAX_CLUSTERS: int = 10
Sequence = str
Sequences = List[str]
IDs = List[str]
SeqToIDs = Dict[Sequence, IDs]
SeqToSeqs = Dict[Sequence, Sequences]
ClusteredIDs = List[IDs]
ClusteredSeqs = List[Sequences]
KmerIDs = Dict[Sequence, int]


def count_distinct_kmers(seqs: Sequences, kmer_size: int) -> KmerIDs:
    """k-mer -> id in first-appearance order over the sequences, positions left to right (reference :26-38; the device
    form is mprg_kmer_dictionary, whose ids are the same)."""
    for seq in seqs:
        if len(seq) < kmer_size:
            raise ValueError(f"Input sequence {seq} has length < kmer size {kmer_size}")
    result: KmerIDs = {}
    for seq in seqs:
        for start in range(len(seq) - kmer_size + 1):
            result.setdefault(seq[start:start + kmer_size], len(result))
    return result


def count_kmer_occurrences(seqs: Sequences, kmers: KmerIDs) -> np.ndarray:
    """Dense count matrix, one row per sequence, columns in id order (reference :41-56; device: mprg_kmer_counts)."""
    kmer_size = len(next(iter(kmers)))
    result = np.zeros((len(seqs), len(kmers)))
    for j, seq in enumerate(seqs):
        for i in range(len(seq) - kmer_size + 1):
            result[j, kmers[seq[i:i + kmer_size]]] += 1
    return result


def get_majority_char_in_column(sequences: Sequences, col_idx: int) -> str:
    max_idx = len(sequences[0]) - 1
    if not (0 <= col_idx <= max_idx):
        raise ValueError(f"Column index {col_idx} not in range(0,{max_idx})")
    return Counter(seq[col_idx] for seq in sequences).most_common(1)[0][0]      # ties: the symbol seen first


def get_majority_string(sequences: Sequences) -> str:
    if len(sequences) == 1:
        return sequences[0]
    seqlen = len(sequences[0])
    if not all(len(seq) == seqlen for seq in sequences):
        raise ValueError("Not all sequences have the same length")
    return "".join(get_majority_char_in_column(sequences, c) for c in range(seqlen))


def hamming_distance(seq1: Sequence, seq2: Sequence) -> int:
    return sum(1 for i in range(len(seq1)) if seq1[i] != seq2[i])


def get_distances(sequences: Sequences, consensus_string: Sequence) -> Iterator[int]:
    return (hamming_distance(seq, consensus_string) for seq in sequences)


def get_one_ref_like_threshold_distance(seqlen: int) -> int:
    return 1 if seqlen < LENGTH_THRESHOLD else int(DISTANCE_THRESHOLD * seqlen)


def _one_reference_like_host(sequences: Sequences) -> bool:
    majority = get_majority_string(sequences)
    threshold = get_one_ref_like_threshold_distance(len(sequences[0]))
    return all(d <= threshold for d in get_distances(sequences, majority))


def _device_applies(clusters: ClusteredSeqs) -> bool:
    """The device kernel sees alignments of the 12-symbol alphabet whose rows are not empty once ungapped; anything else is
    an input the reference answers with its own errors, which the host forms above reproduce.  Inputs of the kernel's domain
    always go to the device: without libmprg_hip.so and a GPU get_backend() raises (no CPU fallback of the hot path)."""
    for seqs in clusters:
        if not seqs or len(seqs[0]) == 0 or any(len(s) != len(seqs[0]) for s in seqs):
            return False
        if any(set(s) - set("ACGT-RYKMSWN") or not s.replace("-", "") for s in seqs):
            return False
    return True


def sequences_are_one_reference_like(sequences: Sequences) -> bool:
    """Every sequence within the threshold Hamming distance of the column-wise majority string (reference :100-104)."""
    if _device_applies([sequences]):
        return not BatchEngine(get_backend(), 5, 1).some_cluster_not_one_reference_like([sequences])[0]
    return _one_reference_like_host(sequences)


def cluster_further(clusters: ClusteredSeqs) -> bool:
    """True if some cluster is not one-reference-like (reference :107-111)."""
    if _device_applies(clusters):
        return any(BatchEngine(get_backend(), 5, 1).some_cluster_not_one_reference_like(clusters))
    return any(not _one_reference_like_host(seqs) for seqs in clusters)


def extract_clusters(seqdict: Union[SeqToIDs, SeqToSeqs], cluster_assignment: List[int]) -> Union[ClusteredIDs, ClusteredSeqs]:
    """Values of `seqdict` grouped by cluster label, clusters in label order (reference :114-133)."""
    value_pool = list(seqdict.values())
    if len(cluster_assignment) != len(value_pool):
        raise ValueError("Mismatch between number of sequences/ID lists and number of cluster assignments")
    num_clusters = max(cluster_assignment) + 1
    if set(cluster_assignment) != set(range(num_clusters)):
        raise ValueError("Inconsistent cluster numbering (likely reason: more input sequences that clustered data points)")
    result: List[list] = [[] for _ in range(num_clusters)]
    for label, elems in zip(cluster_assignment, value_pool):
        result[label].extend(elems)
    return result


@dataclass
class ClusteringResult:
    """reference :136-157."""
    clustered_ids: ClusteredIDs
    sequences: Optional[Sequences] = None

    @property
    def no_clustering(self) -> bool:
        return len(self.clustered_ids) == 1

    @property
    def have_precomputed_sequences(self) -> bool:
        return self.sequences is not None

    def __str__(self):
        return self.__repr__()


def merge_sequences(*seqlists: Sequences, first_seq: str) -> Sequences:
    """first_seq, then every other sequence of the lists in order, IUPAC-expanded and deduplicated (reference :160-191)."""
    found = False
    rest = []
    for seqlist in seqlists:
        for sequence in seqlist:
            if sequence == first_seq:
                found = True
            else:
                rest.append(sequence)
    assert found, f"Provided first sequence argument ({first_seq}) not found in provided list of sequences ({seqlists})"
    return expand_sequences([first_seq] + rest)


def merge_clusters(*clusters: ClusteredIDs, first_id: str) -> ClusteredIDs:
    """All clusters of the given lists, the one holding first_id first with that id leading (reference :194-208)."""
    merged, first_cluster = [], []
    for cluster in chain.from_iterable(clusters):
        if first_id in cluster:
            first_cluster = cluster
        else:
            merged.append(cluster)
    if len(first_cluster) == 0:
        raise ValueError(f"Could not find {first_id} in any cluster")
    first_cluster.remove(first_id)
    first_cluster.insert(0, first_id)
    return [first_cluster] + merged


def kmeans_cluster_seqs(alignment: MSA, kmer_size: int) -> ClusteringResult:
    """reference :211-296."""
    ids, seqs = BatchEngine(get_backend(), 5, kmer_size).cluster(alignment, kmer_size)
    return ClusteringResult(ids, seqs)
