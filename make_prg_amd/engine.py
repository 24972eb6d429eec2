"""Level-synchronous from_msa engine: the Python host drives the recursion, the HIP kernels do the array work.

One `BatchEngine.build(msas)` call builds the recursion trees of MANY alignments together.  All node views of one
recursion level (across every MSA of the batch) go through each kernel in ONE launch, so the number of launches and
host<->device round trips depends on the tree depth, not on the number of alignments.

Reference being replaced: NodeFactory.build and everything below it (make_prg/recursion_tree.py:401-471 with
from_msa/interval_partition.py, from_msa/cluster_sequences.py, utils/seq_utils.py); SURVEY.md §8(a) rows A1-A15.
Host keeps: branch selection (A1), cluster bookkeeping on ids (A12, A14), node numbering (A15), PRG emission (A16).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .backend import MprgError
from .msa import CODE_GAP, MSA, decode, encode

VF, PF = 12, 12
BIT_GAP, BIT_N, BITS_IUPAC = 1 << 4, 1 << 11, 0x7E0
MAX_CLUSTERS = 10   # from_msa/cluster_sequences.py:23
N_INIT = 10         # scikit-learn 1.3.0 default the reference's pinned environment runs with
ROWS_PER_CHUNK = 512
PREPARE_LDS_MAX = 156 * 1024         # MPRG_KMEANS_PREPARE_LDS_MAX (include/mprg.h)
PREPARE_LDS_CLASSES = (12 * 1024, 24 * 1024, 64 * 1024, PREPARE_LDS_MAX)      # one launch per class of LDS need (bytes)
TILE_COLS = 1024
INGEST_TILE = 128                    # MPRG_INGEST_TILE (include/mprg.h): rows x columns of an mprg_ingest work item
IUPAC = {"R": "GA", "Y": "TC", "K": "GT", "M": "AC", "S": "GC", "W": "AT", "A": "A", "C": "C", "G": "G", "T": "T"}


class SequenceCurationError(Exception):
    """utils/seq_utils.py:73-74."""


class PartitioningError(Exception):
    """from_msa/interval_partition.py:12-13."""


def _align(x: int, a: int) -> int:
    return (x + a - 1) // a * a


@dataclass
class NodeRec:
    """One recursion-tree node (host record).  rows: indices into the MSA (None = all rows, in order)."""
    msa: int
    parent: int                 # index into EngineResult.nodes, -1 for root
    level: int                  # nesting level the node is created with
    rows: Optional[np.ndarray]
    col0: int
    ncols: int
    kind: str = "?"             # "leaf" | "interval" | "cluster"
    children: List[int] = field(default_factory=list)
    keep_cols: Optional[np.ndarray] = None    # columns that are not all-gap (device mask), stored-alignment columns
    consensus: Optional[np.ndarray] = None    # per-column consensus code (0..3) or 255
    leaf_rows: Optional[np.ndarray] = None    # leaf: view row positions of the distinct ungapped rows, in order
    node_id: int = -1


@dataclass
class LocusResult:
    index: int
    nodes: List[NodeRec]
    root: int
    error: Optional[Exception] = None
    stats: dict = field(default_factory=dict)


class _LazyCodes:
    """engine.codes[i]: the cell codes of alignment i on the HOST (the device has its own copies) — only tree dumps and the
    rare leaves with ambiguity codes read them, so they are made on first use (bytes outside the alphabet read as 'A',
    as on the device; such loci are dropped anyway)."""

    def __init__(self, msas):
        self._msas, self._cache = msas, {}

    def __len__(self):
        return len(self._msas)

    def __getitem__(self, i):
        i = int(i)
        if i not in self._cache:
            c = encode(self._msas[i].data) if self._msas[i].data.size else np.zeros((0, 0), np.uint8)
            self._cache[i] = np.where(c == 255, 0, c).astype(np.uint8)
        return self._cache[i]


class BatchEngine:
    def __init__(self, backend, max_nesting: int = 5, min_match_length: int = 7):
        self.be = backend
        self.max_nesting = max_nesting
        self.L = min_match_length
        self._uniform_cache: Dict[int, object] = {}
        self.timers: Dict[str, float] = {}
        self.counters: Dict[str, float] = dict(cells_all=0, cells_clustered=0, kmeans_bytes=0, fits=0, levels=0,
                                               launches=0, syncs=0, plan_misses=0, plan_resumes=0)

    # ------------------------------------------------------------------------------------------------ packing
    def _resolve_pending_n(self, msas: List[MSA]):
        """Load-time majority consensus (utils/seq_utils.py:246-290) for the alignments that still hold N: the per-column
        residue counts of ALL of them come from one device launch, the seeded random choice is the host's."""
        from .msa import consensus_from_counts
        idx = [i for i, m in enumerate(msas) if getattr(m, "pending_n", False)]
        if not idx:
            return
        be = self.be
        table = np.zeros((len(idx), 4), np.int64)
        raw_off = col_off = 0
        for j, i in enumerate(idx):
            S, C = msas[i].data.shape
            table[j] = (raw_off, S, C, col_off)
            raw_off += S * C
            col_off += C
        raw = np.concatenate([np.ascontiguousarray(msas[i].data).reshape(-1) for i in idx])
        cnt = (table[:, 2] + 255) // 256
        work = np.empty((int(cnt.sum()), 2), np.int32)
        work[:, 0] = np.repeat(np.arange(len(idx)), cnt)
        work[:, 1] = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        d_raw, d_tab, d_work, d_out = be.upload(raw), be.upload(table), be.upload(work), be.empty(80 * max(col_off, 1))
        be.call("mprg_column_residue_counts", be.ptr(d_raw), be.ptr(d_tab), be.ptr(d_work), len(work), be.ptr(d_out), be.stream,
                work=float(raw.nbytes))
        res = be.download(d_out, np.int32, 20 * col_off).reshape(col_off, 20).astype(np.int64)
        for j, i in enumerate(idx):
            m = msas[i]
            co, C = int(table[j, 3]), int(table[j, 2])
            is_n = m.data == ord("N")
            last = int(np.nonzero(is_n.any(axis=0))[0].max())
            cons = consensus_from_counts(m.data, res[co:co + C, :10], res[co:co + C, 10:], upto=last + 1)
            if not m.data.flags.writeable:
                m.data = m.data.copy()
            m.data[is_n] = np.broadcast_to(cons, m.data.shape)[is_n]
            m.pending_n = False

    def _pack(self, msas: List[MSA]):
        """Lay the batch out in one arena (row-major + transposed coded copy per MSA): the host uploads each alignment
        once, as the parser's ASCII matrix; mprg_ingest codes and transposes on the device."""
        be = self.be
        self._resolve_pending_n(msas)
        self.codes = _LazyCodes(msas)
        self.bad: Dict[int, Exception] = {}
        M = len(msas)
        metas = []
        itab = np.zeros((max(M, 1), 9), np.int64)
        off = raw_off = tile = 0
        for i, m in enumerate(msas):
            S, C = m.data.shape if m.data.ndim == 2 else (0, 0)
            if S == 0:
                C = 0
            pitchC, pitchS = _align(max(C, 1), 16), _align(max(S, 1), 16)
            rm = off
            off = _align(rm + S * pitchC, 256)
            cm = off
            off = _align(cm + C * pitchS, 256)
            metas.append((rm, cm, pitchC, pitchS, S, C))
            itab[i] = (raw_off, S, C, rm, cm, pitchC, pitchS, -1, tile)
            raw_off += S * C
            tile += ((S + INGEST_TILE - 1) // INGEST_TILE) * ((C + INGEST_TILE - 1) // INGEST_TILE)
        self.meta = metas
        parts = [np.ascontiguousarray(m.data).reshape(-1) for m in msas if m.data.size]
        raw = np.concatenate(parts) if parts else np.zeros(1, np.uint8)
        arena_bytes = off + 256
        self.d_arena = be.empty(arena_bytes)
        d_raw, d_itab, d_status = be.upload(raw), be.upload(itab), be.empty(4 * max(M, 1))
        be.call("mprg_ingest", be.ptr(d_raw), be.ptr(d_itab), M, tile, None, be.ptr(self.d_arena), arena_bytes,
                be.ptr(d_status), be.stream, work=3.0 * raw_off)
        status = be.download(d_status, np.int32, M) if M else np.zeros(0, np.int32)
        for i in np.nonzero(status)[0].tolist():
            # any byte outside ACGT-RYKMSWN ends in SequenceCurationError in the reference (it reaches a
            # SequenceExpander check in interval partitioning or leaf emission; utils/seq_utils.py:96-104)
            m = msas[i]
            r, c = np.argwhere(encode(m.data) == 255)[0]
            self.bad[i] = SequenceCurationError(
                f"A slice of a sequence has a disallowed base ({chr(m.data[r, c])!r} in {m.ids[r]}). Redo sequence curation.")
        self.counters["arena_bytes"] = int(arena_bytes)

    # ------------------------------------------------------------------------------------------------ views
    def _view_table(self, nodes: List[NodeRec], idxs: List[int]):
        """int64 [n][VF] view descriptors + the row-index pool for these nodes."""
        n = len(idxs)
        tab = np.zeros((n, VF), np.int64)
        pool, pool_off, cache = [], 0, {}
        col_off = row_off = 0
        for j, ni in enumerate(idxs):
            nd = nodes[ni]
            rm, cm, pitchC, pitchS, S, C = self.meta[nd.msa]
            if nd.rows is None:
                roff, nrows = -1, S
            else:
                key = id(nd.rows)
                if key not in cache:
                    cache[key] = pool_off
                    pool.append(nd.rows)
                    pool_off += len(nd.rows)
                roff, nrows = cache[key], len(nd.rows)
            tab[j] = (rm, cm, pitchC, pitchS, roff, nrows, nd.col0, nd.ncols, col_off, row_off, 0, 0)
            col_off += nd.ncols
            row_off += nrows
        rowidx = np.concatenate(pool).astype(np.int32) if pool else np.zeros(1, np.int32)
        return tab, rowidx, col_off, row_off

    @staticmethod
    def _mask_work(tab: np.ndarray, rows_per_chunk: Optional[int] = None):
        """Work items {view, tile column, first row} of mprg_column_masks, vectorised over the views.
        Returns (work, rows_per_chunk).  The row chunk is chosen so that a launch has >= ~1000 workgroups when the
        level offers that much work (measured on MI355X: a 600 MB view streams at 4.9 TB/s with ~1200 workgroups of
        512 rows x 1024 columns, 3.1 TB/s with 600, 1.2 TB/s with 18 000 tiny ones)."""
        n = tab.shape[0]
        lead = tab[:, 6] % 4                                  # col0 % 4: tiles start 4-aligned in arena columns
        ntile = (tab[:, 7] + lead + TILE_COLS - 1) // TILE_COLS
        if rows_per_chunk is None:
            rows_per_chunk = 64
            for rpc in (1024, 512, 256, 128):
                if int((ntile * ((tab[:, 5] + rpc - 1) // rpc)).sum()) >= 1024:
                    rows_per_chunk = rpc
                    break
        nchunk = (tab[:, 5] + rows_per_chunk - 1) // rows_per_chunk
        per = ntile * nchunk
        total = int(per.sum())
        view = np.repeat(np.arange(n), per)
        start = np.repeat(np.cumsum(per) - per, per)
        k = np.arange(total) - start
        nch = np.repeat(nchunk, per)
        tile = k // nch
        chunk = k % nch
        work = np.empty((total, 3), np.int32)
        work[:, 0] = view
        work[:, 1] = tile * TILE_COLS - np.repeat(lead, per)
        work[:, 2] = chunk * rows_per_chunk
        return work, rows_per_chunk

    @staticmethod
    def _row_chunk_work(tab: np.ndarray, chunk: int = 256) -> np.ndarray:
        """{view, row chunk} pairs covering every row of every view (kernels that walk rows of big views in parallel)."""
        cnt = (tab[:, 5] + chunk - 1) // chunk
        total = int(cnt.sum())
        w = np.empty((total, 2), np.int32)
        w[:, 0] = np.repeat(np.arange(tab.shape[0]), cnt)
        w[:, 1] = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        return w

    @staticmethod
    def _items(cnt: np.ndarray) -> np.ndarray:
        """{view, item} pairs: cnt[v] items of view v."""
        total = int(cnt.sum())
        w = np.empty((total, 2), np.int32)
        w[:, 0] = np.repeat(np.arange(len(cnt)), cnt)
        w[:, 1] = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        return w

    @classmethod
    def _gap_run_work(cls, tab: np.ndarray) -> np.ndarray:
        """Work items of mprg_partition's gap-run launch: (256-row chunk, 512-column segment) pairs of every view (gr_items,
        csrc/k_partition.inc)."""
        return cls._items(((tab[:, 5] + 255) // 256) * ((tab[:, 7] + 511) // 512))

    @classmethod
    def _dedupe_work(cls, tab: np.ndarray) -> np.ndarray:
        """Work items of mprg_ungap_dedupe: row chunks of 256 rows, 8 for a view of more than 4 096 columns (ug_chunks,
        csrc/k_rows.inc)."""
        chunk = np.where(tab[:, 7] > 4096, 8, 256)
        return cls._items((tab[:, 5] + chunk - 1) // chunk)

    # ------------------------------------------------------------------------------------------------ main entry
    def load(self, msas: List[MSA]):
        """Ingest: encode + lay out + upload the batch; afterwards the alignments are resident in HBM."""
        self._ids = [m.ids for m in msas]
        self._msas = msas
        self._pack(msas)
        self.be.synchronize()

    def build(self, msas: List[MSA]) -> List[LocusResult]:
        self.load(msas)
        return self.run()

    def run(self, root_level=0, root_is_tree_root=True) -> List[LocusResult]:
        """The hot path on the resident batch: the whole recursion forest, level by level.
        root_level / root_is_tree_root (one value, or one per alignment): re-entry of NodeFactory.build below an
        existing parent (LeafNode._update_leaf, recursion_tree.py:374-376) starts at the parent's nesting level and does
        not force a MultiIntervalNode."""
        be = self.be
        msas = self._msas
        per = lambda v: list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * len(msas)
        root_levels, self._root_forced = per(root_level), per(root_is_tree_root)
        assert len(root_levels) == len(msas) and len(self._root_forced) == len(msas)
        nodes: List[NodeRec] = []
        results = [LocusResult(i, nodes, -1) for i in range(len(msas))]
        frontier: List[int] = []
        for i, m in enumerate(msas):
            if i in self.bad:
                results[i].error = self.bad[i]
                continue
            S, C = self.meta[i][4], self.meta[i][5]
            nodes.append(NodeRec(i, -1, int(root_levels[i]), None, 0, C))
            results[i].root = len(nodes) - 1
            frontier.append(len(nodes) - 1)
        failed = set(self.bad)
        while frontier:
            frontier = [ni for ni in frontier if nodes[ni].msa not in failed]
            if not frontier:
                break
            self.counters["levels"] += 1
            frontier = self._level(nodes, frontier, results, failed)
        for r in results:
            if r.error is None and r.root >= 0:
                self._number_nodes(nodes, r.root)
        self.nodes = nodes
        return results

    # ------------------------------------------------------------------------------------------------ one level
    def _masks_and_partition(self, nodes, frontier, given_mask: Optional[np.ndarray] = None, L: Optional[int] = None):
        """K1 + K2 over the views of `frontier`.  given_mask: use these per-column presence masks instead of
        computing them (IntervalPartitioner called with an explicit consensus string)."""
        be = self.be
        L = self.L if L is None else L
        tab, rowidx, total_cols, total_rows = self._view_table(nodes, frontier)
        n = len(frontier)
        cells = float((tab[:, 5] * tab[:, 7]).sum())
        d_views, d_rowidx = be.upload(tab), be.upload(rowidx)
        if given_mask is None:
            work, rpc = self._mask_work(tab)
            d_work = be.upload(work)
            d_mask = be.zeros(4 * total_cols)
            be.call("mprg_column_masks", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work),
                    work.shape[0], rpc, be.ptr(d_mask), be.stream, work=cells)
        else:
            d_mask = be.upload(given_mask.astype(np.uint32))
        d_maxrun, d_stack, d_ivflag = be.zeros(4 * total_cols), be.empty(16 * total_cols), be.zeros(8 * total_cols)
        d_iv, d_niv, d_status = be.empty(12 * total_cols), be.empty(4 * n), be.empty(4 * n)
        wr = self._gap_run_work(tab)
        d_wr = be.upload(wr)
        be.call("mprg_partition", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(d_rowidx), n, be.ptr(d_mask), L,
                be.ptr(d_wr), len(wr), be.ptr(d_maxrun), be.ptr(d_stack), be.ptr(d_ivflag), be.ptr(d_iv), be.ptr(d_niv),
                be.ptr(d_status), None, None, None, None, 0, None, 0, be.stream, work=cells)
        self.counters["launches"] += 2
        mask = be.download(d_mask, np.uint32, total_cols)
        n_iv = be.download(d_niv, np.int32, n)
        status = be.download(d_status, np.int32, n)
        iv = be.download(d_iv, np.int32, 3 * total_cols).reshape(-1, 3).copy()
        iv[:, 2] &= 1                     # bit 1 (MPRG_IV_PURE) is a hint for the array-at-a-time host
        return tab, d_views, d_rowidx, mask, n_iv, iv, status

    def _level(self, nodes, frontier, results, failed) -> List[int]:
        tab, d_views, d_rowidx, mask, n_iv, iv, status = self._masks_and_partition(nodes, frontier)
        n = len(frontier)
        total_cols = int(tab[:, 7].sum())
        self.counters["cells_all"] += int((tab[:, 5] * tab[:, 7]).sum())

        # consensus codes for all columns of the level (utils/seq_utils.py:228-238), vectorised
        m = mask & ~np.uint32(BIT_N)
        single = (m != 0) & ((m & (m - 1)) == 0) & ((m & BITS_IUPAC) == 0) & (m != BIT_GAP)
        cons = np.full(total_cols, 255, np.uint8)
        cons[single] = np.log2(m[single]).astype(np.uint8)
        allgap = mask == BIT_GAP

        next_frontier: List[int] = []
        cluster_cands: List[int] = []      # positions j in frontier
        dedupe_leaves: List[int] = []      # leaves whose rows must be grouped on the device
        for j, ni in enumerate(frontier):
            nd = nodes[ni]
            if nd.msa in failed:
                continue
            co = int(tab[j, 8])
            nd.keep_cols = ~allgap[co:co + nd.ncols]
            nd.consensus = cons[co:co + nd.ncols]
            if status[j]:
                err = (SequenceCurationError("All sequences in this slice contained N. Redo sequence curation.")
                       if status[j] & 2 else PartitioningError("Failed interval partitioning"))
                results[nd.msa].error = err
                failed.add(nd.msa)
                continue
            k = int(n_iv[j])
            ivs = iv[co:co + k]
            if k == 1 and ivs[0, 2] == 0:
                nd.kind = "leaf"
                if (nd.consensus == 255).any():
                    dedupe_leaves.append(j)
                else:
                    nd.leaf_rows = None       # single sequence == the consensus string
            elif k > 1 or (nd.parent < 0 and self._root_forced[nd.msa]):
                nd.kind = "interval"
                for a, b, _t in ivs:
                    nodes.append(NodeRec(nd.msa, ni, nd.level, nd.rows, nd.col0 + int(a), int(b) - int(a) + 1))
                    nd.children.append(len(nodes) - 1)
                    next_frontier.append(len(nodes) - 1)
            else:
                cluster_cands.append(j)
        if cluster_cands or dedupe_leaves:
            next_frontier.extend(self._cluster_stage(nodes, frontier, tab, d_views, d_rowidx, cluster_cands,
                                                     dedupe_leaves, results, failed))
        return next_frontier


    def _cluster_further(self, d_sub, d_rowidx, sub, act_tab, k, d_dor, d_labels, d_assign, d_scratch, d_further, d_gcodes=None):
        """mprg_cluster_further for the problems of act_tab (rows of the problem table); returns bool per problem.
        (A one-workgroup form with the view's cells in LDS was built and measured in round 2: 10.3 ms against 7.4 ms per
        3 000 alignments — its 58 KB of LDS halve the residency of a kernel that is bound by instruction issue — dropped.)"""
        launch = self._cluster_further_plan(d_sub, d_rowidx, sub, act_tab, k, d_dor, d_labels, d_assign, d_scratch, d_further,
                                            d_gcodes=d_gcodes)
        launch(None)
        return self.be.download(d_further, np.int32, len(act_tab)).astype(bool)

    @staticmethod
    def _cluster_further_items(sub, act_tab):
        """Work lists of mprg_cluster_further for the problems of act_tab: (problem, column tile), (problem, row chunk), and
        the members' cells (the algorithmic bytes of one evaluated k)."""
        nA = len(act_tab)
        views = act_tab[:, 0]

        def items(cnt):
            w = np.empty((int(cnt.sum()), 2), np.int32)
            w[:, 0] = np.repeat(np.arange(nA), cnt)
            w[:, 1] = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            return w

        # (column tiles: 256 columns, 32 for a problem of more than 1 024 rows — cf_col_tiles, csrc/k_cluster.inc)
        tile = np.where(sub[views, 5] > 1024, 32, 256)
        chunk = np.where((sub[views, 7] >= 512) & (sub[views, 5] > 1024), 16, 256)          # (row chunks: 256 rows, 16 for a problem of >= 512 columns and > 1 024 rows — cf_row_chunks)
        return items((sub[views, 7] + tile - 1) // tile), items((sub[views, 5] + chunk - 1) // chunk), float((sub[views, 5] * sub[views, 7]).sum())

    def _cluster_further_plan(self, d_sub, d_rowidx, sub, act_tab, k, d_dor, d_labels, d_assign, d_scratch, d_further, staged=None,
                              d_gcodes=None):
        """Uploads the work lists of mprg_cluster_further now and returns launch(d_km_info): the caller can put a KMeans
        launch between the two (an upload waits for the stream's earlier kernels) and fetch d_further when it pleases.
        staged: (device act_tab, device column items, n, device row items, n, work) if the caller has uploaded them already."""
        be = self.be
        nA = len(act_tab)
        if staged is None:
            wc, wr, work = self._cluster_further_items(sub, act_tab)
            staged = (be.upload(act_tab), be.upload(wc), len(wc), be.upload(wr), len(wr), work)
        d_sp, d_wc, n_wc, d_wr, n_wr, work = staged

        def launch(d_km_info):
            be.call("mprg_cluster_further", be.ptr(self.d_arena), be.ptr(d_sub), be.ptr(d_rowidx), be.ptr(d_sp), nA, k,
                    be.ptr(d_dor), be.ptr(d_labels) if k > 1 else None, be.ptr(d_assign) if (k > 1 and d_assign is not None) else None,
                    be.ptr(d_wc), n_wc, be.ptr(d_wr), n_wr, be.ptr(d_scratch), be.ptr(d_further),
                    be.ptr(d_km_info) if d_km_info is not None else None, be.ptr(d_gcodes) if d_gcodes is not None else None,
                    None, be.stream, work=work)
            self.counters["launches"] += 2
        return launch

    def _kmeans_prepare(self, d_ptab, D, V, d_x, d_ws):
        """mprg_kmeans_prepare with the problems split by the LDS their matrix needs (include/mprg.h): every workgroup of a
        launch allocates the launch's lds_bytes, so the small problems (the rule) go in a launch of their own and keep
        several workgroups per CU resident; matrices beyond the budget take the global-memory form."""
        be = self.be
        need = 8 * (D * (V | 1) + 2 * V)
        work = float((8 * D * V).sum())
        lo = -1
        for hi in PREPARE_LDS_CLASSES:
            idx = np.nonzero((need > lo) & (need <= hi))[0].astype(np.int32)
            lo = hi
            if len(idx):
                d_idx = be.upload(idx)
                be.call("mprg_kmeans_prepare", be.ptr(d_ptab), len(idx), be.ptr(d_x), be.ptr(d_ws), be.ptr(d_idx), len(idx),
                        int(need[idx].max()), 0, 0, be.stream, work=work)
                work = 0.0
        idx = np.nonzero(need > PREPARE_LDS_MAX)[0].astype(np.int32)
        if len(idx):
            d_idx = be.upload(idx)
            be.call("mprg_kmeans_prepare", be.ptr(d_ptab), len(idx), be.ptr(d_x), be.ptr(d_ws), 0, 0, 0, be.ptr(d_idx), len(idx),
                    be.stream, work=work)

    def _dedupe(self, d_sub, d_rowidx, n_views: int, tot_rows: int, tot_u: int, work: float = 0.0, sub=None, wr=None):
        """mprg_ungap_dedupe over the views of `d_sub`; returns the device buffers by name.  wr: (device list of {view, row
        chunk} items, their number) if the caller built them on the device, else they are made here from the host table `sub`."""
        be = self.be
        R = max(tot_rows, 1)
        b = dict(ucodes=be.empty(tot_u), gcodes=be.empty(tot_u), hash=be.empty(16 * R), ulen=be.empty(4 * R), rep_u=be.empty(4 * R),
                 rep_g=be.empty(4 * R), d_of_row=be.empty(4 * R), s_of_row=be.empty(4 * R), reps_pos=be.empty(4 * R),
                 reps_len=be.empty(4 * R), seqrow=be.empty(4 * R), occ_off=be.empty(8 * (R + n_views)),
                 summary=be.empty(64 * max(n_views, 1)))
        if wr is None:
            w = self._dedupe_work(sub)
            wr = (be.upload(w), len(w))
        d_wr, n_wr = wr
        be.call("mprg_ungap_dedupe", be.ptr(self.d_arena), be.ptr(d_sub), be.ptr(d_rowidx), n_views, self.L,
                be.ptr(d_wr), n_wr, be.ptr(b["ucodes"]), be.ptr(b["hash"]), be.ptr(b["ulen"]), be.ptr(b["rep_u"]), be.ptr(b["rep_g"]),
                be.ptr(b["d_of_row"]), be.ptr(b["s_of_row"]), be.ptr(b["reps_pos"]), be.ptr(b["reps_len"]),
                be.ptr(b["seqrow"]), be.ptr(b["occ_off"]), be.ptr(b["summary"]), be.ptr(b["gcodes"]), be.stream, work=work)
        self.counters["launches"] += 2
        return b

    # ------------------------------------------------------------------------------------------------ clustering
    def _cluster_stage(self, nodes, frontier, tab, d_views, d_rowidx, cands, leaves, results, failed) -> List[int]:
        """A9-A14 for every single-non-match-interval view of the level (+ row grouping for non-trivial leaves)."""
        be, K = self.be, self.L
        sel = cands + leaves
        # sub-table of the selected views with their own row offsets and ucodes regions
        sub = tab[sel].copy()
        usize = sub[:, 5] * ((sub[:, 7] + 15) // 16 * 16)          # ungapped rows, row-major, 16-byte pitch
        sub[:, 10] = np.cumsum(usize) - usize
        sub[:, 9] = np.cumsum(sub[:, 5]) - sub[:, 5]
        sub[:, 8] = np.cumsum(sub[:, 7]) - sub[:, 7]
        tot_rows, tot_u, tot_cols = int(sub[:, 5].sum()), int(usize.sum()), int(sub[:, 7].sum())
        d_sub = be.upload(sub)
        dd = self._dedupe(d_sub, d_rowidx, len(sel), tot_rows, tot_u, work=2.0 * float((sub[:, 5] * sub[:, 7]).sum()), sub=sub)
        d_ucodes, d_ulen = dd["ucodes"], dd["ulen"]
        ulen = be.download(dd["ulen"], np.int32, tot_rows)
        rep_u = be.download(dd["rep_u"], np.int32, tot_rows)
        rep_g = be.download(dd["rep_g"], np.int32, tot_rows)

        new_nodes: List[int] = []
        for q in range(len(cands), len(sel)):           # leaves: distinct ungapped rows in first-appearance order
            nd = nodes[frontier[sel[q]]]
            ro, S = int(sub[q, 9]), int(sub[q, 5])
            nd.leaf_rows = np.nonzero(rep_u[ro:ro + S] == np.arange(S))[0]

        # ---- per candidate: groups, early exits (A13, D <= 2), problem list
        probs = []      # dicts
        for q in range(len(cands)):
            j = sel[q]
            ni = frontier[j]
            nd = nodes[ni]
            ro, S = int(sub[q, 9]), int(sub[q, 5])
            self.counters["cells_clustered"] += S * nd.ncols
            ru, rg, ul = rep_u[ro:ro + S], rep_g[ro:ro + S], ulen[ro:ro + S]
            is_rep = ru == np.arange(S)
            nd.leaf_rows = np.nonzero(is_rep)[0]
            n_unique_u = int(is_rep.sum())
            n_unique_g = int((rg == np.arange(S)).sum())
            # recursion_tree.py:538-556 / :475-494 — the clustering result is only used if all of these allow it
            if nd.level + 1 >= self.max_nesting or n_unique_u <= 2 or n_unique_u < n_unique_g:
                nd.kind = "leaf"
                continue
            long_reps = np.nonzero(is_rep & (ul >= K))[0]
            D = len(long_reps)
            if D <= 2:                                   # cluster_sequences.py:235-246: single cluster
                nd.kind = "leaf"
                continue
            occ = (ul[long_reps] - K + 1).astype(np.int64)
            probs.append(dict(q=q, ni=ni, nd=nd, S=S, ro=ro, ru=ru, ul=ul, long_reps=long_reps, D=D,
                              occ_off=np.concatenate(([0], np.cumsum(occ))), T=int(occ.sum())))
        if probs:
            self._run_kmeans_problems(nodes, probs, sub, d_sub, d_rowidx, d_ucodes, d_ulen, tot_rows, tot_cols, dd["d_of_row"], dd["gcodes"])
            for p in probs:
                new_nodes.extend(self._finish_cluster_node(nodes, p))
        return new_nodes

    def _uniforms_all(self):
        """Device buffer with numpy RandomState(2).random_sample streams of every k = 2..10, and each k's offset."""
        if "all" not in self._uniform_cache:
            parts, offs, o = [], {}, 0
            for k in range(2, MAX_CLUSTERS + 1):
                n = N_INIT * (1 + (k - 1) * (2 + int(np.log(k))))
                parts.append(self.be.random_sample(2, n))
                offs[k] = o
                o += n
            self._uniform_cache["all"] = (self.be.upload(np.concatenate(parts)), offs)
        return self._uniform_cache["all"]

    def _run_kmeans_problems(self, nodes, probs, sub, d_sub, d_rowidx, d_ucodes, d_ulen, tot_rows, tot_cols, d_dor, d_gcodes):
        be, K = self.be, self.L
        P = len(probs)
        ptab = np.zeros((P, PF), np.int64)
        seqrow, occ_offs = [], []
        so = oo = to = fo = 0
        for i, p in enumerate(probs):
            cap = 16
            while cap < 2 * p["T"]:
                cap *= 2
            ptab[i, 0], ptab[i, 1], ptab[i, 2], ptab[i, 3] = p["q"], p["D"], so, p["T"]
            ptab[i, 4], ptab[i, 5], ptab[i, 6], ptab[i, 11] = to, cap, oo, fo
            seqrow.append(p["long_reps"].astype(np.int32))
            occ_offs.append(p["occ_off"].astype(np.int64))
            so += p["D"]
            oo += p["D"] + 1
            to += 16 * cap
            fo += _align(p["T"], 16)
        d_seqrow, d_occ = be.upload(np.concatenate(seqrow)), be.upload(np.concatenate(occ_offs))
        d_table, d_flag, d_V = be.empty(to), be.empty(fo), be.empty(4 * P)
        d_ptab = be.upload(ptab)
        be.call("mprg_kmer_dictionary", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(d_ucodes), be.ptr(d_ulen),
                be.ptr(d_seqrow), be.ptr(d_occ), be.ptr(d_table), be.ptr(d_flag), be.ptr(d_V), be.stream)
        V = be.download(d_V, np.int32, P).astype(np.int64)
        if ((V >> 24) == 0x7f).any():
            raise MprgError("k-mer dictionary: no hash seed separated the k-mers of a clustering problem (k-mer size > 16)")
        ptab[:, 5] |= (V >> 24) << 40               # k > 16: the hash seed that held rides in the capacity field (mprg_kmer_counts)
        V &= 0xFFFFFF
        xo = wo = lo = 0
        for i, p in enumerate(probs):
            p["V"] = int(V[i])
            ptab[i, 7], ptab[i, 8], ptab[i, 9], ptab[i, 10] = V[i], xo, wo, lo
            xo += p["D"] * int(V[i])
            need = int(be.lib.mprg_kmeans_workspace_doubles(p["D"], int(V[i]), MAX_CLUSTERS, N_INIT))
            if need < 0:
                raise MprgError("a k-mer count matrix has more than 4 194 304 features: beyond the KMeans kernels' pairwise-sum stack")
            wo += need
            lo += p["D"]
        d_ptab = be.upload(ptab)
        d_x, d_ws = be.zeros(8 * xo), be.empty(8 * wo)
        d_labels, d_kmst, d_info = be.empty(4 * lo), be.zeros(4 * P), be.empty(64 * P)
        be.call("mprg_kmer_counts", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(d_ucodes), be.ptr(d_ulen),
                be.ptr(d_seqrow), be.ptr(d_occ), be.ptr(d_table), be.ptr(d_x), be.stream)
        self._kmeans_prepare(d_ptab, ptab[:, 1], ptab[:, 7], d_x, d_ws)
        self.counters["launches"] += 3

        for p in probs:
            p["assign"] = np.zeros(p["D"], np.int64)
            p["num_clusters"] = 1
        d_scratch, d_further = be.empty(12 * tot_cols + 64), be.empty(4 * P)

        def check(active_idx, k):
            """cluster_further() for the listed problems on the labels the KMeans select step just wrote."""
            return self._cluster_further(d_sub, d_rowidx, sub, ptab[active_idx], k, d_dor, d_labels, None, d_scratch,
                                         d_further)

        # cluster_sequences.py:256-274 — `while cluster_further(...): num_clusters += 1; KMeans(num_clusters).fit(X).predict(X)` with its
        # accept / revert / stop rules — is the LIBRARY's loop (mprg_cluster_loop: the problem's workgroup walks k = 2..10 itself; the
        # same kernels and the same control as the batch host's forests, forest.py).  Here: the k = 1 check, then one call.
        fur = check(list(range(P)), 1)
        if fur.any():
            d_uni, offs = self._uniforms_all()
            uoffs = np.zeros(MAX_CLUSTERS + 1, np.int32)
            for k_, o_ in offs.items():
                uoffs[k_] = o_
            d_numcl, d_active = be.upload(np.ones(P, np.int32)), be.upload(fur.astype(np.int32))
            d_assign, d_stats = be.zeros(4 * lo), be.zeros(8 * 96)
            args = (be.ptr(d_sub), be.ptr(d_ptab), P, N_INIT, be.ptr(d_uni), uoffs.ctypes.data, be.ptr(d_x), be.ptr(d_ws), be.ptr(d_dor),
                    be.ptr(d_gcodes), be.ptr(d_scratch), be.ptr(d_labels), be.ptr(d_assign), be.ptr(d_info), be.ptr(d_kmst), be.ptr(d_numcl),
                    be.ptr(d_active), be.ptr(d_stats))
            # (the LDS form's classes first, then the general form for the rounds no class holds: include/mprg.h MPRG_LOOP_*)
            be.call("mprg_cluster_loop", *args, 16, be.stream)
            be.call("mprg_cluster_loop", *args, 1, be.stream)
            self.counters["launches"] += 7
            stats = be.download(d_stats, np.int64, 96)
            if stats[82]:
                raise MprgError("KMeans empty-cluster relocation: the selection ran out of frames (more than 5^10 samples in a fit)")
            numcl = be.download(d_numcl, np.int32, P)
            assign_all = be.download(d_assign, np.int32, lo)
            self.counters["fits"] += int(stats[80])
            self.counters["kmeans_bytes"] += float(stats[85:86].view(np.float64)[0]) + float(stats[93:94].view(np.float64)[0])
            for i, p in enumerate(probs):
                if not fur[i]:
                    continue
                p["num_clusters"] = int(numcl[i])
                p["assign"] = assign_all[int(ptab[i, 10]):int(ptab[i, 10]) + p["D"]].astype(np.int64)

    def _finish_cluster_node(self, nodes, p) -> List[int]:
        """cluster_sequences.py:276-296 + recursion_tree.py:457-469, :558-572 on ids."""
        nd: NodeRec = p["nd"]
        D, k = p["D"], p["num_clusters"]
        if k == 1 or k == D:
            nd.kind = "leaf"
            return []
        m = nd.msa
        view_rows = np.arange(self.meta[m][4]) if nd.rows is None else nd.rows
        ids = self._ids[m]
        row_ids = [ids[r] for r in view_rows]
        S, ru, ul = p["S"], p["ru"], p["ul"]
        clusters: List[List[str]] = [[] for _ in range(int(p["assign"].max()) + 1)]
        if set(p["assign"].tolist()) != set(range(len(clusters))):
            raise ValueError("Inconsistent cluster numbering")
        rows_of_rep: Dict[int, List[int]] = {}
        for i in range(S):
            rows_of_rep.setdefault(int(ru[i]), []).append(i)
        for d, rep in enumerate(p["long_reps"].tolist()):
            clusters[int(p["assign"][d])].extend(row_ids[i] for i in rows_of_rep[rep])
        short_reps = [r for r in sorted(rows_of_rep) if ul[r] < self.L]
        clusters.extend([[row_ids[i] for i in rows_of_rep[r]] for r in short_reps])
        first_id = row_ids[0]
        chosen = next((c for c in clusters if first_id in c), None)
        if chosen is None:
            raise ValueError(f"Could not find {first_id} in any cluster")
        rest = [c for c in clusters if c is not chosen]
        chosen = list(chosen)
        chosen.remove(first_id)
        clusters = [[first_id] + chosen] + rest
        assert sum(len(c) for c in clusters) == S, "Each input sequence should be in a cluster"
        nd.kind = "cluster"
        nd.level += 1                     # recursion_tree.py:459: the node itself sits one nesting level down
        out = []
        for c in clusters:
            cset = set(c)
            sel = np.asarray([view_rows[i] for i in range(S) if row_ids[i] in cset], dtype=np.int32)
            nodes.append(NodeRec(m, p["ni"], nd.level, sel, nd.col0, nd.ncols))
            nd.children.append(len(nodes) - 1)
            out.append(len(nodes) - 1)
        return out

    # ------------------------------------------------------------------------------------------------ numbering
    @staticmethod
    def _number_nodes(nodes: List[NodeRec], root: int):
        """Node ids in the reference's construction order: preorder (recursion_tree.py:48-55)."""
        nid = 0
        stack = [root]
        while stack:
            ni = stack.pop()
            nodes[ni].node_id = nid
            nid += 1
            stack.extend(reversed(nodes[ni].children))


# ----------------------------------------------------------------------------------------------------- emission
def expand_sequences(seqs: List[str]) -> List[str]:
    """utils/seq_utils.py:116-153 (order-preserving dedupe, drop rows with N, IUPAC expansion, dedupe)."""
    import itertools
    allowed = set(IUPAC) | {"N"}
    for s in seqs:
        if not set(s) <= allowed:
            raise SequenceCurationError(f"A slice of a sequence has a disallowed base.\nSequence: {s}\nRedo sequence curation.\n")
    out, seen, seen_in = [], set(), set()
    for s in seqs:
        if s in seen_in:
            continue
        seen_in.add(s)
        if "N" in s:
            continue
        if set(s) <= set("ACGT"):
            if s not in seen:
                seen.add(s)
                out.append(s)
            continue
        for combo in itertools.product(*(IUPAC[b] for b in s)):
            e = "".join(combo)
            if e not in seen:
                seen.add(e)
                out.append(e)
    if not out:
        raise SequenceCurationError(f"All sequences in this slice contained N. Redo sequence curation.\nSequences: {seqs}")
    return out


def leaf_sequences(engine: BatchEngine, nd: NodeRec) -> List[str]:
    """The alleles a leaf emits (recursion_tree.py:272-274): distinct ungapped rows in order, expanded."""
    if nd.leaf_rows is None:
        return [decode(nd.consensus).tobytes().decode()]
    codes = engine.codes[nd.msa]
    rows = np.arange(codes.shape[0]) if nd.rows is None else nd.rows
    block = codes[rows[nd.leaf_rows], nd.col0:nd.col0 + nd.ncols]
    seqs = []
    for r in block:
        seqs.append(decode(r[r != CODE_GAP]).tobytes().decode())
    return expand_sequences(seqs)


def build_prg(engine: BatchEngine, res: LocusResult):
    """PrgBuilder.build_prg (prg_builder.py:100-105) + the three traversals (recursion_tree.py:194-300).
    Returns (prg string, {(start, end): node index})."""
    nodes = res.nodes
    out: List[str] = []
    pos = 0
    index: Dict[Tuple[int, int], int] = {}
    site = 5
    # iterative preorder with explicit "after child" actions
    stack: List[tuple] = [("node", res.root)]
    while stack:
        item = stack.pop()
        if item[0] == "text":
            out.append(item[1])
            pos += len(item[1])
            continue
        nd = nodes[item[1]]
        if nd.kind == "interval":
            for c in reversed(nd.children):
                stack.append(("node", c))
        elif nd.kind == "cluster":
            s = site
            site += 2
            out.append(f" {s} ")
            pos += len(out[-1])
            nch = len(nd.children)
            for i in reversed(range(nch)):
                stack.append(("text", f" {s + 1 if i < nch - 1 else s} "))
                stack.append(("node", nd.children[i]))
        else:
            seqs = leaf_sequences(engine, nd)
            if len(seqs) == 1:
                out.append(seqs[0])
                index[(pos, pos + len(seqs[0]))] = item[1]
                pos += len(seqs[0])
            else:
                s = site
                site += 2
                t = f" {s} "
                out.append(t)
                pos += len(t)
                for i, q in enumerate(seqs):
                    out.append(q)
                    index[(pos, pos + len(q))] = item[1]
                    pos += len(q)
                    t = f" {s + 1 if i < len(seqs) - 1 else s} "
                    out.append(t)
                    pos += len(t)
    return "".join(out), index, site


def stored_alignment(engine: BatchEngine, nd: NodeRec, ids: List[str]):
    """node.alignment of the reference: the node's sub-alignment without its all-gap columns
    (recursion_tree.py:45 -> utils/seq_utils.py:193-216).  Column selection = the device's all-gap mask."""
    codes = engine.codes[nd.msa]
    rows = np.arange(codes.shape[0]) if nd.rows is None else nd.rows
    block = codes[rows, nd.col0:nd.col0 + nd.ncols][:, nd.keep_cols]
    return [ids[r] for r in rows], decode(block)


def tree_dump(engine: BatchEngine, res: LocusResult, ids: List[str]) -> list:
    """Flat preorder dump (same shape as oracle.from_msa_oracle.tree_dump) for tree-equality checks."""
    out = []
    nodes = res.nodes
    stack = [res.root]
    while stack:
        ni = stack.pop()
        nd = nodes[ni]
        rids, block = stored_alignment(engine, nd, ids)
        out.append(dict(id=nd.node_id, kind=nd.kind, level=nd.level,
                        parent=None if nd.parent < 0 else nodes[nd.parent].node_id,
                        rows=[[i, r.tobytes().decode()] for i, r in zip(rids, block)],
                        children=[nodes[c].node_id for c in nd.children]))
        stack.extend(reversed(nd.children))
    return out


# ----------------------------------------------------------------------------------------------------- single-view API
def _one_view_engine(be, alignment: MSA, max_nesting=5, L=7) -> Tuple[BatchEngine, List[NodeRec]]:
    eng = BatchEngine(be, max_nesting, L)
    eng.load([alignment])
    if 0 in eng.bad:
        raise eng.bad[0]
    S, C = eng.meta[0][4], eng.meta[0][5]
    return eng, [NodeRec(0, -1, 0, None, 0, C)]


def _bm_column_masks(self: BatchEngine, alignment: MSA) -> np.ndarray:
    eng, nodes = _one_view_engine(self.be, alignment)
    tab, rowidx, total_cols, _ = eng._view_table(nodes, [0])
    be = eng.be
    d_views, d_rowidx = be.upload(tab), be.upload(rowidx)
    work, rpc = eng._mask_work(tab)
    d_work, d_mask = be.upload(work), be.zeros(4 * total_cols)
    be.call("mprg_column_masks", be.ptr(eng.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work), work.shape[0],
            rpc, be.ptr(d_mask), be.stream)
    return be.download(d_mask, np.uint32, total_cols)


def _bm_compact_columns(self: BatchEngine, alignment: MSA) -> np.ndarray:
    """The alignment's cell codes without its all-gap columns (A8), by mprg_column_masks + mprg_compact_columns."""
    eng, nodes = _one_view_engine(self.be, alignment)
    tab, rowidx, total_cols, _ = eng._view_table(nodes, [0])
    be = eng.be
    S, C = int(tab[0, 5]), int(tab[0, 7])
    d_views, d_rowidx = be.upload(tab), be.upload(rowidx)
    work, rpc = eng._mask_work(tab)
    d_work, d_mask = be.upload(work), be.zeros(4 * total_cols)
    be.call("mprg_column_masks", be.ptr(eng.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work), work.shape[0],
            rpc, be.ptr(d_mask), be.stream)
    wr = eng._row_chunk_work(tab, 64)
    d_wr, d_out, d_off, d_kept = be.upload(wr), be.empty(S * C), be.upload(np.zeros(1, np.int64)), be.zeros(4)
    be.call("mprg_compact_columns", be.ptr(eng.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_wr), len(wr), 64,
            be.ptr(d_mask), be.ptr(d_out), be.ptr(d_off), be.ptr(d_kept), be.stream)
    kept = int(be.download(d_kept, np.int32, 1)[0])
    return be.download(d_out, np.uint8, S * kept).reshape(S, kept)


def _bm_partition(self: BatchEngine, alignment: MSA, L: int, consensus: Optional[str] = None):
    """IntervalPartitioner on one alignment.  With `consensus` given, the partition follows THAT string (the
    reference's constructor takes it as an argument); '*' columns are non-match, anything else match."""
    if len(alignment) == 0:
        # reference tests partition bare consensus strings with an empty alignment: no rows, no probes
        n = len(consensus or "")
        fake = MSA.from_strings(["A" * n]) if n else None
        if fake is None:
            return []
        eng, nodes = _one_view_engine(self.be, fake, L=L)
        eng.meta[0] = eng.meta[0][:4] + (0, eng.meta[0][5])        # zero rows
    else:
        eng, nodes = _one_view_engine(self.be, alignment, L=L)
    given = None
    if consensus is not None:
        arr = np.frombuffer(consensus.encode(), np.uint8)
        given = np.where(arr == ord("*"), np.uint32(0b10011), np.uint32(1))   # '*': {A, C, -}; else a single base
    tab, _, _, mask, n_iv, iv, status = eng._masks_and_partition(nodes, [0], given_mask=given, L=L)
    if status[0] & 2:
        raise SequenceCurationError("All sequences in this slice contained N. Redo sequence curation.")
    if status[0] & 1:
        raise PartitioningError("Failed interval partitioning")
    return [(int(a), int(b), int(t)) for a, b, t in iv[:int(n_iv[0])]]


def _bm_row_groups(self: BatchEngine, alignment: MSA):
    eng, nodes = _one_view_engine(self.be, alignment)
    be = eng.be
    tab, rowidx, total_cols, total_rows = eng._view_table(nodes, [0])
    d_rowidx = be.upload(rowidx)
    S = int(tab[0, 5])
    tab[0, 10] = 0
    d_sub = be.upload(tab)
    dd = eng._dedupe(d_sub, d_rowidx, 1, S, S * ((int(tab[0, 7]) + 15) // 16 * 16), sub=tab)
    ru, rg = be.download(dd["rep_u"], np.int32, S), be.download(dd["rep_g"], np.int32, S)
    ulen = be.download(dd["ulen"], np.int32, S)
    ar = np.arange(S)
    return int((ru == ar).sum()), int((rg == ar).sum()), ru, ulen


def _bm_cluster(self: BatchEngine, alignment: MSA, kmer_size: int):
    """kmeans_cluster_seqs (cluster_sequences.py:211-296) on one alignment → (clustered_ids, sequences|None)."""
    eng, nodes = _one_view_engine(self.be, alignment, max_nesting=1 << 30, L=kmer_size)
    nodes[0].level = 0
    be, K = eng.be, kmer_size
    tab, rowidx, total_cols, total_rows = eng._view_table(nodes, [0])
    d_rowidx = be.upload(rowidx)
    S = int(tab[0, 5])
    sub = tab.copy()
    sub[0, 10] = 0
    d_sub = be.upload(sub)
    dd = eng._dedupe(d_sub, d_rowidx, 1, S, S * ((int(sub[0, 7]) + 15) // 16 * 16), sub=sub)
    d_ucodes, d_ulen = dd["ucodes"], dd["ulen"]
    ul, ru = be.download(dd["ulen"], np.int32, S), be.download(dd["rep_u"], np.int32, S)
    is_rep = ru == np.arange(S)
    long_reps = np.nonzero(is_rep & (ul >= K))[0]
    short_reps = np.nonzero(is_rep & (ul < K))[0]
    D = len(long_reps)
    ids = alignment.ids
    rows = alignment.rows_as_strings()
    rows_of_rep: Dict[int, List[int]] = {}
    for i in range(S):
        rows_of_rep.setdefault(int(ru[i]), []).append(i)

    def ungapped(i):
        return rows[i].replace("-", "")

    def promote(clusters):
        first_id = ids[0]
        chosen = next((c for c in clusters if first_id in c), None)
        if chosen is None:
            raise ValueError(f"Could not find {first_id} in any cluster")
        rest = [c for c in clusters if c is not chosen]
        chosen = list(chosen)
        chosen.remove(first_id)
        return [[first_id] + chosen] + rest

    def single():
        everything = [ids[i] for r in long_reps for i in rows_of_rep[int(r)]] + \
                     [ids[i] for r in short_reps for i in rows_of_rep[int(r)]]
        first_seq = ungapped(0)
        others = [ungapped(int(r)) for r in list(long_reps) + list(short_reps) if ungapped(int(r)) != first_seq]
        return promote([everything]), expand_sequences([first_seq] + others)

    if D <= 2:
        return single()
    occ = (ul[long_reps] - K + 1).astype(np.int64)
    p = dict(q=0, ni=0, nd=nodes[0], S=S, ro=0, ru=ru, ul=ul, long_reps=long_reps, D=D,
             occ_off=np.concatenate(([0], np.cumsum(occ))), T=int(occ.sum()))
    eng._run_kmeans_problems(nodes, [p], sub, d_sub, d_rowidx, d_ucodes, d_ulen, S, int(sub[0, 7]), dd["d_of_row"], dd["gcodes"])
    k = p["num_clusters"]
    if k == 1 or k == D:
        return single()
    clusters: List[List[str]] = [[] for _ in range(int(p["assign"].max()) + 1)]
    if set(p["assign"].tolist()) != set(range(len(clusters))):
        raise ValueError("Inconsistent cluster numbering")
    for d, rep in enumerate(long_reps.tolist()):
        clusters[int(p["assign"][d])].extend(ids[i] for i in rows_of_rep[rep])
    clusters.extend([[ids[i] for i in rows_of_rep[int(r)]] for r in short_reps])
    clusters = promote(clusters)
    assert sum(len(c) for c in clusters) == S, "Each input sequence should be in a cluster"
    return clusters, None


def _bm_not_one_reference_like(self: BatchEngine, clusters: List[List[str]]) -> List[bool]:
    """cluster_sequences.py:100-111 for a list of clusters (each a list of equal-length gapped rows): True where some row
    is further than the one-reference-like threshold from its cluster's majority string.  One alignment per cluster, all of
    them through mprg_ungap_dedupe + mprg_cluster_further (k = 1) in one batch."""
    be = self.be
    msas = [MSA.from_strings(c) for c in clusters]
    eng = BatchEngine(be, 1 << 30, 1)
    eng.load(msas)
    for i in eng.bad:
        raise eng.bad[i]
    nodes = [NodeRec(i, -1, 0, None, 0, eng.meta[i][5]) for i in range(len(msas))]
    tab, rowidx, total_cols, total_rows = eng._view_table(nodes, list(range(len(nodes))))
    d_rowidx = be.upload(rowidx)
    sub = tab.copy()
    usize = sub[:, 5] * ((sub[:, 7] + 15) // 16 * 16)
    sub[:, 10] = np.cumsum(usize) - usize
    d_sub = be.upload(sub)
    dd = eng._dedupe(d_sub, d_rowidx, len(nodes), total_rows, int(usize.sum()), sub=sub)
    sm = be.download(dd["summary"], np.int64, 8 * len(nodes)).reshape(len(nodes), 8)
    act = np.zeros((len(nodes), PF), np.int64)
    act[:, 0], act[:, 1] = np.arange(len(nodes)), sm[:, 2]
    d_scratch, d_further = be.empty(12 * total_cols + 64), be.empty(4 * len(nodes))
    return eng._cluster_further(d_sub, d_rowidx, sub, act, 1, dd["d_of_row"], None, None, d_scratch, d_further).tolist()


BatchEngine.some_cluster_not_one_reference_like = _bm_not_one_reference_like
BatchEngine.column_masks = _bm_column_masks
BatchEngine.compact_columns = _bm_compact_columns
BatchEngine.partition = _bm_partition
BatchEngine.row_groups = _bm_row_groups
BatchEngine.cluster = _bm_cluster
