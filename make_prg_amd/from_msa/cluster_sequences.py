"""Sequence clustering of a sub-alignment — API of make_prg/from_msa/cluster_sequences.py.  The k-mer
featurisation, KMeans and the one-reference-like test run on the device (mprg_kmer_*, mprg_kmeans_*,
mprg_cluster_further); the id bookkeeping stays on the host."""
from dataclasses import dataclass
from typing import List, Optional

from ..device import get_backend
from ..engine import BatchEngine
from ..msa import MSA

DISTANCE_THRESHOLD: float = 0.2
LENGTH_THRESHOLD: int = 5
MAX_CLUSTERS: int = 10
IDs = List[str]
ClusteredIDs = List[IDs]


@dataclass
class ClusteringResult:
    clustered_ids: ClusteredIDs
    sequences: Optional[List[str]] = None

    @property
    def no_clustering(self) -> bool:
        return len(self.clustered_ids) == 1

    @property
    def have_precomputed_sequences(self) -> bool:
        return self.sequences is not None


def get_one_ref_like_threshold_distance(seqlen: int) -> int:
    return 1 if seqlen < LENGTH_THRESHOLD else int(DISTANCE_THRESHOLD * seqlen)


def kmeans_cluster_seqs(alignment: MSA, kmer_size: int) -> ClusteringResult:
    """reference :211-296."""
    ids, seqs = BatchEngine(get_backend(), 5, kmer_size).cluster(alignment, kmer_size)
    return ClusteringResult(ids, seqs)
