"""Host of the batched from_msa build (the throughput path of bench.py and the CLI): the Python host drives the recursion,
every table lives on the device.

Rounds 1-2 kept per-NODE NumPy arrays here and made every decision of a recursion level on the host (one host process kept
the device busy a sixth of the time).  Now the node table, the view / problem / work-item tables of every launch and the state
of the reference's `k = 2, 3, ...` clustering loop are device arrays built by the kernels of csrc/k_forest.inc
(mprg_forest_* in include/mprg.h): each step is "count per item -> exclusive prefix sums -> fill", and this host reads one
small header of totals per step — it owns the buffers (PyTorch-ROCm) and must size the next ones — and nothing else.  The
clustering loop runs without any host decision: before round k the device settles round k-1 (accepted / reverted / done)
for every problem and retires the finished ones (their workgroups of the round's launches return at once).
PRG assembly (preorder ids, site numbers, text offsets, allele copies, markers) also runs over the device node table.

Reference semantics: recursion_tree.py:401-471 (NodeFactory.build), cluster_sequences.py:211-296,
prg_builder.py:100-119 + recursion_tree.py:194-300 (traversals).  Row ids are assumed unique inside an alignment
(the reference partitions cluster children by id); the per-alignment API (engine.BatchEngine) keeps id semantics.
"""
from typing import Dict, List, Optional

import os

import numpy as np

from .backend import MprgError
from .engine import (MAX_CLUSTERS, N_INIT, PF, VF, BatchEngine, PartitioningError, SequenceCurationError, expand_sequences)
from .msa import CODE_GAP, decode

KIND_LEAF, KIND_INTERVAL, KIND_CLUSTER = 0, 1, 2
FUSED_VIEWS = os.environ.get("MPRG_FUSED_VIEWS", "1") != "0"     # fused small-view launch shape of mprg_partition
_ACGT = np.frombuffer(b"ACGT-RYKMSWN????", dtype=np.uint8)

# include/mprg.h
NODE_FIELDS, ASM_FIELDS, VC, HDR = 16, 8, 16, 96
(N_MSA, N_PARENT, N_LEVEL, N_ROWS_OFF, N_NROWS, N_COL0, N_NCOLS, N_FLAGS, N_KIND, N_FIRST_CHILD, N_NCHILD, N_LVL, N_REPS_OFF,
 N_NSEQ, N_ACHARS, N_AUX) = range(16)
NF_PURE, NF_SPECIAL, NF_FORCED = 1, 2, 4
A_SIZE, A_PRE, A_SITE, A_TOTAL, A_START, A_NSEQ, A_ACHARS, A_JOB = range(8)
_F_NAMES = ("NODES N_NODES META N_MSAS FAILED ERR_FIRST POOL POOL_USED ARENA MAX_NESTING MIN_MATCH FUSED_ENABLED N_INIT VALS "
            "SCAN_TMP HDR F0 N LVL VIEWS VIEW2NODE FUSED_LIST OTHER_LIST MASK_WORK RPC_IDX GAP_WORK VIEW_OUT IV_PACKED N_VIEWS SUB "
            "SELNODE DD_WORK SUMMARY NSEL T1 WORK_COLS WORK_ROWS FURTHER NPQ PTAB0 PTAB DV P CLS_LISTS NUM_CLUSTERS ACTIVE KINFO "
            "KM_INFO KM_STATUS SPT SP SPLITNODE CHILD_SIZES NSPLITS ASM ROOT_OF SPECIAL_LIST SPECIAL_CAP PATCH N_PATCH LEVELS "
            "N_LEVELS VALS_MSA VALS_NODE VALS_POS N_SITES JOBS OUT MSA_BASE UOFF").split()
FI = {name: i for i, name in enumerate(_F_NAMES)}
FI["HDR_HOST"], FI["FIT_LISTS"], FI["KM_MODE"], FI["INDEX_OUT"], FI["EX_RECORDS"], FI["EX_ROWS"] = 80, 81, 82, 83, 84, 85
# mprg_forest_level (a level without a host wait): device state, buffers the per-step host passes to the data entry points, capacities
for _i, _n in enumerate("DS LEVEL_INDEX MASK MAXRUN STACK IVFLAG IV NIV STATUS IVC UCODES GCODES HASHES ULEN REP_U REP_G D_OF_ROW S_OF_ROW REPS_POS "
                        "REPS_LEN SEQROW OCC_OFF CF_SCRATCH TABLE FLAG X WS LABELS ASSIGN UNIFORMS UOFF_HOST LOOP_FORMS CAP".split()):
    FI[_n] = 96 + _i
(CAP_TCOLS, CAP_NFUSED, CAP_NOTHER, CAP_ITEMS, CAP_NGAP, CAP_NODES, CAP_SROWS, CAP_UBYTES, CAP_SCOLS, CAP_NDD, CAP_WC, CAP_WR, CAP_TABLE, CAP_FLAG,
 CAP_LO, CAP_XD, CAP_WSD) = range(17)
CAP_CLS, CAP_LDS, CAP_NCHILD, CAP_POOL, CAP_BIG = 17, 22, 26, 27, 28
STEP_SIZES_SHAPE, STEP_BEGIN = 7, 9          # overflow codes beside the six steps (MPRG_STEP_*)
FI["SIDE_STREAM"] = 160
FI["MAX_ROWS"] = 161
DS_OVERFLOW, DS_F0, DS_N, DS_NNODES, DS_POOL_USED, DS_LEVEL, DS_NFAILED, DS_GLOBAL, DS_LEVEL_WORDS = 0, 1, 2, 3, 4, 5, 6, 16, 6 * 96
# levels without host waits once the engine has a plan (the previous forest of the same resident batch): MPRG_SPECULATIVE=0 keeps the
# per-step host
SPECULATIVE = os.environ.get("MPRG_SPECULATIVE", "1") != "0"
# launch lists of a KMeans round (hdr 86..92): wave form by LDS class, general workgroup form, small workgroup form
# which forms small fits take: bit 0 wave form (measured slower on MI355X: profiles/r03/kmeans_forms.md), bit 1 small workgroups,
# bit 2 (round 6, default) the LDS form for every fit it has a class for — the restarts' state in LDS (csrc/k_kmeans_lds.inc);
# its classes use the wave form's list slots
KM_MODE = int(os.environ.get("MPRG_KM_MODE", "6"))
KM_LDS_ENTRY = "mprg_kmeans_fit_lds" if (KM_MODE & 4) else "mprg_kmeans_fit_wave"
KM_LISTS = ((KM_LDS_ENTRY, 0), (KM_LDS_ENTRY, 1), (KM_LDS_ENTRY, 2), (KM_LDS_ENTRY, 3),
            ("mprg_kmeans_fit", None), ((KM_LDS_ENTRY, 4) if (KM_MODE & 4) else ("mprg_kmeans_fit_small", 0)),
            ((KM_LDS_ENTRY, 5) if (KM_MODE & 4) else ("mprg_kmeans_fit_small", 1)))
# a round's launch lists side by side on side streams: measured flat on MI355X (352 vs 354 ms per forest of 30 000 alignments,
# profiles/r03/kmeans_forms.md), off by default
KM_SIDE_STREAMS = os.environ.get("MPRG_KM_SIDE_STREAMS", "0") != "0"
# a launch list of at most this many fits goes through the SPLIT form (a workgroup per restart + a selection launch,
# mprg_kmeans_fit_split): such a launch lasts one fit latency whatever it holds (profiles/r03/kmeans_split.md); 0 = never
KM_SPLIT_BELOW = int(os.environ.get("MPRG_KM_SPLIT_BELOW", "0"))
# BIG clustering problems (a count matrix of at least this many bytes: hundreds of distinct sequences x thousands of k-mers — what one
# deep alignment's levels hold, BASELINE config D's stress): their level runs the per-round loop and the general-form fits of its rounds
# take a wide workgroup per RESTART (mprg_kmeans_fit_wide) — every phase of such a fit is thousands of chains as long as the k-mer
# dictionary, ten restarts side by side in one workgroup queue behind one CU (profiles/r04/deep_alignment.md); 0 = never
KM_BIG_BYTES = int(os.environ.get("MPRG_KM_BIG_BYTES", str(1 << 20)))
# ... and from this size on the level's big problems are prepared WITHOUT the sample-sample tables of the seeding (mprg_kmeans_prepare_big, with_tables = 0:
# 2.5 D^2 chains per problem, 32 D^2 bytes); the wide fits then compute the few dozen rows they ask for themselves.  Until the tables were made
# by tiles (k_kmeans_prepare_tables_tiled, round 5) this was 160 MB (~1 500 sequences x 16 384 k-mers); with them the tables pay for
# themselves at every size measured (10 000 x 20 000 hierarchical: 0.46 s for the tables, 0.58 s less in the fits), so only problems whose
# tables would take tens of GB go without
KM_NO_TABLES_BYTES = int(os.environ.get("MPRG_KM_NO_TABLES_BYTES", str(4 << 30)))
# ... and a big level of at most this many problems fits EVERY round's general-form KMeans at once (_kloop_rounds, spec_k): a round is
# ten wide workgroups per problem — with five problems fifty CUs of 256, nine rounds one after the other; 0 = never
KM_SPEC_PROBLEMS = int(os.environ.get("MPRG_KM_SPEC_PROBLEMS", "64"))
# k-mer dictionaries by KD_PARTS workgroups per problem when a level's problems hold this many k-mer occurrences on average (0 = never)
KD_PARTS_FROM = int(os.environ.get("MPRG_KD_PARTS_FROM", str(1 << 17)))
KD_PARTS = 128
# the clustering loop: "fused" = a problem's workgroup walks k = 2..10 itself (mprg_cluster_loop; one launch per workgroup form and
# level), "rounds" = one set of launches per round k (the shape of rounds 1-3)
# "auto" (default): fused below KLOOP_ROUNDS_FROM alignments in the engine, rounds from there on.  Measured on MI355X (profiles/r04/
# loop_forms.md): the fused loop wins where launches and waits decide (3 750 alignments per step, one worker: 68.2 k against 65.8 k
# alignments/s; 940: 25.5 against 29.8 ms per pass) and lets a forest be enqueued without waits; with several worker processes sharing
# a saturated device and >= 7 500 alignments each the per-round kernels carry ~4 % less scratch traffic (95.2 k against 91.6 k)
KLOOP = os.environ.get("MPRG_KLOOP", "auto")
KLOOP_ROUNDS_FROM = int(os.environ.get("MPRG_KLOOP_ROUNDS_FROM", "6000"))
LOOP_GENERAL, LOOP_SMALL = "mprg_cluster_loop[general]", "mprg_cluster_loop[small]"          # names the fused launches are timed under
F_FIELDS = 192
PREPARE_CLASSES = 4                      # LDS classes of mprg_kmeans_prepare (+ the global-memory form)
PREPARE_LDS_BOUNDS = (12 * 1024, 24 * 1024, 64 * 1024, 156 * 1024)          # kf_prepare_class (csrc/k_forest.inc), MPRG_KMEANS_PREPARE_LDS_MAX
# ---- capacities of a FIRST-SEEN batch (ForestEngine._caps_predicted): which total of a level — (step, column) of its headers — is a sum
# over which count; [0][14] (the frontier) follows from the level before
_CAP_COLS = {(0, 0): (0, 14), (0, 1): (0, 0), (0, 3): (0, 14), (0, 4): (0, 14), (0, 5): (0, 4), (0, 6): (0, 4), (0, 7): (0, 4), (0, 8): (0, 4),
             (0, 9): (0, 4), (0, 10): (0, 4),
             (1, 0): (0, 0), (1, 1): (0, 0), (1, 2): (1, 1), (1, 3): (1, 1), (1, 4): (1, 1), (1, 5): (1, 1),
             (2, 0): (1, 1), (2, 1): (2, 0), (2, 2): (2, 0),
             (3, 0): (2, 0), (3, 1): (3, 0), (3, 2): (3, 0), (3, 3): (3, 0),
             (4, 0): (3, 0), (4, 1): (3, 0), (4, 2): (3, 0), (4, 3): (3, 0), (4, 4): (3, 0), (4, 5): (3, 0), (4, 6): (3, 0),
             (5, 0): (3, 0), (5, 1): (5, 0), (5, 2): (5, 0)}
_CAP_BY_STEP = {s_: ([i for i, (st, _) in enumerate(_CAP_COLS) if st == s_], [c for (st, c) in _CAP_COLS if st == s_]) for s_ in range(6)}
PLAN_HEAD = float(os.environ.get("MPRG_PLAN_HEAD", "1.35"))          # headroom of a predicted total ...
PLAN_SPREAD = float(os.environ.get("MPRG_PLAN_SPREAD", "24"))        # ... + this / sqrt(items behind it)
PLAN_SLACK_LEVELS = int(os.environ.get("MPRG_PLAN_SLACK_LEVELS", "0"))
PLAN_FLOOR = float(os.environ.get("MPRG_PLAN_FLOOR", "8"))           # room for this many more items of the donor's largest average size
PLAN_TRACE = os.environ.get("MPRG_PLAN_TRACE", "") not in ("", "0")
DONOR_MIN_ROOTS = int(os.environ.get("MPRG_DONOR_MIN_ROOTS", "16"))  # a forest of fewer alignments is nobody's donor
DONOR_MAX_RATIO = 32.0
SPEC_RETRIES = int(os.environ.get("MPRG_SPEC_RETRIES", "6"))         # levels enqueued again after an overflow before the per-step host takes over
SPEC_SPARE_LEVELS = 4                    # room in the device state for levels beyond the capacities' (a forest deeper than its donor's)


class ForestEngine(BatchEngine):
    """load() as BatchEngine; run_forest() builds every tree of the batch; assemble_prgs() emits the PRG strings."""

    @property
    def kloop_fused(self) -> bool:
        """Which form the clustering loop takes for this engine's batch (KLOOP above)."""
        return KLOOP == "fused" or (KLOOP != "rounds" and len(self._msas) < KLOOP_ROUNDS_FROM)

    # ------------------------------------------------------------------------------------------------ plumbing
    def _set(self, **kw):
        """Forest state fields (MPRG_F_*): buffers by address (and kept alive until the level ends), numbers as they are."""
        for name, v in kw.items():
            if isinstance(v, (int, np.integer)):
                self.F[FI[name]] = int(v)
            elif v is None:
                self.F[FI[name]] = 0
            else:
                self.F[FI[name]] = self.be.ptr(v)
                self._alive[name] = v

    def _step(self, name, *extra, n_hdr=0, work=0.0):
        """One mprg_forest_* entry point; returns the first n_hdr words of the step's header (a wait for the device)."""
        self.be.call("mprg_forest_" + name, self.F.ctypes.data, *extra, self.be.stream, work=work)
        self.counters["launches"] += 1
        if n_hdr:          # the device has published the header to pinned host memory: wait for the stream, read
            self.counters["syncs"] = self.counters.get("syncs", 0) + 1
            self.be.synchronize()
            return self._hdr_host[:n_hdr].copy()
        return None

    def _plan_note(self, rec, step: int, h):
        """The per-step host writes down every step's totals: the next forest of this batch is sized from them (run_forest)."""
        if rec is None:
            rec = {s_: np.zeros(HDR, np.int64) for s_ in range(6)}
            rec["rpc_idx"] = 4
            self._plan_rec.append(rec)
        rec[step][:len(h)] = h
        return rec

    def _scratch(self, n_items: int):
        """vals / scan scratch of a count step over n_items items."""
        be = self.be
        self._set(VALS=be.empty(8 * VC * max(n_items, 1)), SCAN_TMP=be.empty(8 * VC * (n_items // 2048 + 2)))

    def _grow_nodes(self, need: int):
        if need > self.cap_nodes:
            new_cap = max(2 * self.cap_nodes, need)
            self.d_nodes = self.be.grown(self.d_nodes, 8 * NODE_FIELDS * self.n_nodes, 8 * NODE_FIELDS * new_cap)
            self.cap_nodes = new_cap
            self._set(NODES=self.d_nodes)

    def _pool_reserve(self, extra_rows: int):
        need = 4 * (self.pool_used + extra_rows)
        if need > self.pool_cap:
            new_cap = max(2 * self.pool_cap, need, 1 << 16)
            self.d_pool = self.be.grown(self.d_pool, 4 * self.pool_used, new_cap)
            self.pool_cap = new_cap
            self._set(POOL=self.d_pool)

    def pool_host(self) -> np.ndarray:
        return self.be.download(self.d_pool, np.int32, self.pool_used).astype(np.int64)

    # ------------------------------------------------------------------------------------------------ forest
    def run_forest(self, root_level=0, root_is_tree_root=True):
        """The whole recursion forest of the resident batch, level by level.  root_level / root_is_tree_root (one value or
        one per alignment): re-entry below an existing parent (LeafNode._update_leaf, recursion_tree.py:374-376) starts at
        the parent's nesting level and does not force a MultiIntervalNode.
        Two ways through a level: the per-step host (`_forest_level`: every step's totals are read back to size the next buffers,
        ~6 waits per level + one per KMeans round) and mprg_forest_level: the level's buffers sized from CAPACITIES, exact counts
        on the device, no wait until the whole forest is enqueued.  Capacities come from the totals of an earlier forest of the
        same resident batch (`_plan`: exact, no headroom) or — a batch seen for the FIRST time, what the command line's chunks
        and a rank's shard are — from the totals of ANOTHER batch (`plan_donor`, see plan_export), scaled by the batches' cells,
        with headroom.  A total beyond its capacity stops the device at that level; the host then enqueues that level and the
        ones after it again with larger buffers (mprg_forest_state_rewind: the levels before it are kept), and leaves the rest
        to the per-step host when that does not settle it or the level holds a BIG clustering problem."""
        self.forest_enqueue(root_level, root_is_tree_root)
        self.forest_finish()

    def forest_enqueue(self, root_level=0, root_is_tree_root=True):
        """First half of run_forest.  With capacities (a plan of its own or a donor's): the whole forest is ENQUEUED and the call
        returns (no wait; several engines on streams of their own can be fed by one host thread in turn); without: the per-step
        host runs the forest here.  forest_finish() completes either."""
        M = len(self._msas)
        per = lambda v, dt: np.asarray(v, dt) if isinstance(v, (list, tuple, np.ndarray)) else np.full(M, v, dt)
        root_levels, forced = per(root_level, np.int64), per(root_is_tree_root, bool)
        key = (root_levels.tobytes(), forced.tobytes())
        self._forest_begin(root_levels, forced, key)
        self._roots_args = (root_levels, forced, key)
        plan = getattr(self, "_plan", None)
        self._pending = None
        caps = None
        if SPECULATIVE and self.be.profile is None and len(self.ok) and self.kloop_fused:
            if plan is not None and plan["key"] == key:
                caps = self._caps_own(plan)
            elif not root_levels.any() and forced.all():          # (whole alignments from their roots: what a donor's totals describe)
                donor = getattr(self, "plan_donor", None) or getattr(self.be, "plan_donor", None)
                caps = self._caps_predicted(donor) if donor is not None else None
        self._whole_roots = not root_levels.any() and bool(forced.all())
        if caps is not None:
            self._pending = self._spec_enqueue(caps)
        else:
            self._forest_exact()
            self._publish_donor()          # (the forest is complete: the NEXT engine's first batch — a pipeline's second chunk, begun before this
            #                                 one's forest_finish — is already sized from it instead of taking the per-step host as well)

    def forest_finish(self):
        """Second half of run_forest: waits for a forest that was enqueued from capacities and looks at the device state (a total
        that did not fit: see run_forest)."""
        by_device = False
        if self._pending is not None:
            by_device = self._spec_finish(self._pending)
            self._pending = None
        self._nodes_hint, self._pool_hint = self.n_nodes + (self.n_nodes >> 4), self.pool_used + (self.pool_used >> 4)
        self._forest_end(check_failed=not by_device or self._spec_failed)
        self._publish_donor()

    def _publish_donor(self):
        """The next batch built on this backend — the command line's next chunk, a rank's next shard — is sized from this one."""
        if self._whole_roots and len(self.ok) >= DONOR_MIN_ROOTS:
            donor = self.plan_export()
            if donor is not None:
                self.be.plan_donor = donor

    def plan_export(self):
        """What another engine needs to size ITS first forest from this engine's last one (`other.plan_donor = this.plan_export()`):
        every step's totals per level, the roots and their cells.  None when the last forest left no plan (a batch with BIG clustering
        problems, or the per-round clustering loop)."""
        plan = getattr(self, "_plan", None)
        if plan is None or not len(self.ok):
            return None
        return dict(levels=plan["levels"], n_roots=len(self.ok), cells=float((self.meta_arr[self.ok, 4] * self.meta_arr[self.ok, 5]).sum()),
                    settings=(self.max_nesting, self.L))

    def _forest_exact(self):
        """The per-step host: every step's totals read back; writes the plan the next forest of this batch is sized from."""
        self._plan_rec = []
        self._big_seen = False
        self._forest_exact_levels(0, len(self.ok))

    def _forest_exact_levels(self, f0: int, n: int):
        while n:
            self.counters["levels"] += 1
            f0, n = self._forest_level(f0, n)
            self._alive = {k: v for k, v in self._alive.items() if k in ("NODES", "META", "FAILED", "ERR_FIRST", "POOL", "ARENA", "HDR", "HDR_HOST", "UNIFORMS")}
        # (a plan is the sizes of a forest whose clustering loops are the fused ones: none for a batch with BIG problems)
        self._plan = dict(key=self._roots_args[2], levels=self._plan_rec, n_nodes=self.n_nodes, pool_used=self.pool_used) \
            if (self.kloop_fused and not self._big_seen) else None

    def _forest_begin(self, root_levels, forced, key):
        """Node table with the roots, row pool, per-locus flags, state fields.  What depends only on the resident batch (roots,
        meta, initial flags) is uploaded once per batch and copied device-to-device afterwards: no wait for the stream here."""
        be = self.be
        M = len(self._msas)
        const = getattr(self, "_dev_const", None)
        if const is None or const["key"] != key:
            meta = np.asarray(self.meta, dtype=np.int64).reshape(M, 6)
            failed = np.zeros(M, bool)
            for i in self.bad:
                failed[i] = True
            ok = np.nonzero(~failed)[0]
            roots = np.zeros((len(ok), NODE_FIELDS), np.int64)
            roots[:, N_MSA], roots[:, N_PARENT], roots[:, N_LEVEL], roots[:, N_ROWS_OFF] = ok, -1, root_levels[ok], -1
            roots[:, N_NROWS], roots[:, N_NCOLS], roots[:, N_FLAGS] = meta[ok, 4], meta[ok, 5], np.where(forced[ok], NF_FORCED, 0)
            roots[:, N_FIRST_CHILD], roots[:, N_LVL], roots[:, N_REPS_OFF], roots[:, N_NSEQ] = -1, -1, -1, 1
            roots[:, N_ACHARS], roots[:, N_AUX] = meta[ok, 5], -1
            root_of = np.full(M, -1, np.int64)
            root_of[ok] = np.arange(len(ok))
            const = self._dev_const = dict(key=key, meta=meta, failed=failed, ok=ok, root_of=root_of, roots_bytes=roots.nbytes,
                                           d_roots=be.upload(roots) if len(ok) else None, d_meta=be.upload(meta) if M else be.zeros(16),
                                           d_failed=be.upload(failed.astype(np.int32)) if M else be.zeros(16))
        self.meta_arr, self.ok, self.root_of = const["meta"], const["ok"], const["root_of"]
        ok = self.ok
        self.failed = const["failed"].copy()
        self.errors: Dict[int, Exception] = dict(self.bad)
        self.F = np.zeros(F_FIELDS, np.int64)
        self._alive: Dict[str, object] = {}
        # capacity: what the previous forest of this engine needed (a resident batch is rebuilt step after step), else a guess
        # of 64 nodes per alignment; growing means a device-to-device copy of the table
        self.n_nodes = len(ok)
        self.cap_nodes = max(getattr(self, "_nodes_hint", 0), 64 * len(ok), 1024) if getattr(self, "cap_floor", 0) == 0 else \
            max(getattr(self, "_nodes_hint", 0), len(ok), self.cap_floor)
        self.d_nodes = be.grown(const["d_roots"], const["roots_bytes"], 8 * NODE_FIELDS * self.cap_nodes) if len(ok) else \
            be.empty(8 * NODE_FIELDS * self.cap_nodes)
        self.pool_cap, self.pool_used = 4 * getattr(self, "_pool_hint", 0), 0
        self.d_pool = be.empty(max(self.pool_cap, 16))
        self.d_hdr = be.zeros(8 * HDR)
        if not hasattr(self, "_hdr_buf"):
            self._hdr_buf, raw = be.host_visible(8 * HDR)
            self._hdr_host = raw.view(np.int64)
        self.d_failed = be.grown(const["d_failed"], 4 * max(M, 1), 4 * max(M, 4))
        self.d_err = be.full(8 * max(M, 1), 255)
        self.d_meta = const["d_meta"]
        self.levels: List[dict] = []
        self._tab = None
        self._spec_failed = False
        d_uni, uoff = self._uniforms_all()
        self._set(NODES=self.d_nodes, N_NODES=self.n_nodes, META=self.d_meta, N_MSAS=M, FAILED=self.d_failed, ERR_FIRST=self.d_err,
                  POOL=self.d_pool, POOL_USED=0, ARENA=self.d_arena, MAX_NESTING=self.max_nesting, MIN_MATCH=self.L,
                  FUSED_ENABLED=int(FUSED_VIEWS), N_INIT=N_INIT, HDR=self.d_hdr, HDR_HOST=self._hdr_buf)
        for k_, o_ in uoff.items():
            self.F[FI["UOFF"] + k_] = o_
        self.F[FI["MAX_ROWS"]] = int(self.meta_arr[ok, 4].max()) if len(ok) else 1          # no view has more rows than its root
        self._d_uni, self._uoff = d_uni, uoff

    def _forest_end(self, check_failed: bool):
        """Per-locus policy: the locus is dropped, the batch goes on; the first failing view in frontier order names the error."""
        be = self.be
        M = len(self._msas)
        if M and len(self.ok) and check_failed:
            failed_dev = be.download(self.d_failed, np.int32, M) != 0
            new = np.nonzero(failed_dev & ~self.failed)[0]
            if len(new):
                err = be.download(self.d_err, np.uint64, M)
                for mi in new.tolist():
                    code = int(err[mi]) & 3
                    if code == 3:
                        raise MprgError("mprg_partition was given a view for its fused launch shape that does not fit it")
                    self.failed[mi] = True
                    self.errors[mi] = (SequenceCurationError("All sequences in this slice contained N. Redo sequence curation.")
                                       if code == 2 else PartitioningError("Failed interval partitioning"))

    # ------------------------------------------------------------------------------------------------ levels without host waits
    def _caps_own(self, plan):
        """Capacities = the totals of this batch's previous forest (no headroom: the same forest again)."""
        return dict(levels=[{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in lv.items()} for lv in plan["levels"]],
                    n_nodes=int(plan["n_nodes"]), pool=int(plan["pool_used"]), predicted=False)

    def _caps_predicted(self, donor):
        """Capacities of a batch seen for the first time, from the totals of ANOTHER batch: every total scaled by the ratio of the
        batches' cells, times a headroom that grows as the number of items behind the total shrinks (a total over n items is
        predicted to ~ 1 / sqrt(n)), plus room for PLAN_FLOOR more items of the largest average size any level of the donor saw.
        The chain of frontiers is made consistent (a level's
        frontier holds what the level before may create), the LDS classes are launched with their upper bounds."""
        if donor.get("settings") != (self.max_nesting, self.L):
            return None
        ok = self.ok
        r = float((self.meta_arr[ok, 4] * self.meta_arr[ok, 5]).sum()) / max(float(donor["cells"]), 1.0)
        if not (1.0 / DONOR_MAX_RATIO <= r <= DONOR_MAX_RATIO):          # (a batch of another order of magnitude says little)
            return None
        # (everything that depends on the donor alone is kept with it: a pipeline sizes every chunk from the same donor, and a 3 750-alignment
        #  pass is 50 ms — the 1.5 ms of Python loops this used to take were half of what a first pass lost against a planned one)
        pc = donor.get("_pred")
        if pc is None:
            src = donor["levels"]
            cols = list(_CAP_COLS)
            # (a pipeline's donor is the chunk before: used once — so this, too, is array work: the levels' step blocks stacked per step)
            M = [np.stack([np.asarray(lv[s_], np.float64) for lv in src]) if len(src) else np.zeros((0, HDR)) for s_ in range(6)]
            A = np.stack([M[st][:, c] for (st, c) in cols], axis=1) if len(src) else np.zeros((0, len(cols)))
            B = np.stack([M[_CAP_COLS[k][0]][:, _CAP_COLS[k][1]] for k in cols], axis=1) if len(src) else np.zeros((0, len(cols)))
            unit = np.maximum(1.0, (A / np.maximum(B, 1.0)).max(axis=0)) if len(src) else np.ones(len(cols))
            items = M[0][:, 5:10].copy()
            by_step = _CAP_BY_STEP
            pc = donor["_pred"] = dict(A=A, B=B, unit=unit, items=items, by_step=by_step)
        A, B, unit, by_step = pc["A"], pc["B"], pc["unit"], pc["by_step"]
        n_src = A.shape[0]
        pred, n_pred = r * A, r * B
        # the items that effectively make up a total: the counted ones — or fewer: a count of RARE items (the problems of one LDS class
        # among thousands: 12 predicted, 30 seen) scatters like its own value, a total of a few big items like their number
        n_eff = np.minimum(n_pred, pred / unit)
        caps_all = np.ceil(pred * (PLAN_HEAD + PLAN_SPREAD / np.sqrt(n_eff + 1.0)) + PLAN_FLOOR * unit).astype(np.int64)
        caps_none = np.ceil(PLAN_FLOOR * unit).astype(np.int64)          # (a level beyond the donor's: the floor's room only)
        rpc_all = r * pc["items"]

        levels, n_front = [], len(ok)
        n_nodes, pool = len(ok), 0
        # (PLAN_SLACK_LEVELS levels more than the donor had, with the floor's room only: 0 — a batch that nests deeper than its donor
        #  stops at the level that must find an empty frontier and is enqueued again from there; every level costs ~60 launches)
        for li in range(n_src + PLAN_SLACK_LEVELS):
            row = caps_all[li] if li < n_src else caps_none
            out = {s_: np.zeros(HDR, np.int64) for s_ in range(6)}
            for s_, (idx, cs_) in by_step.items():
                if idx:
                    out[s_][cs_] = row[idx]
            out[0][14] = n_front
            for c in (0, 3, 4):          # views, fused views, other views: at most the frontier
                out[0][c] = min(int(out[0][c]), n_front)
            items = rpc_all[li] if li < n_src else np.zeros(5)
            out["rpc_idx"] = next((i for i in range(4) if items[i] >= 1024), 4)
            for c in range(PREPARE_CLASSES):
                out[4][16 + c] = PREPARE_LDS_BOUNDS[c]
            levels.append(out)
            n_front = int(out[1][0] + out[5][2])
            n_nodes += n_front
            pool += int(out[5][1])
        return dict(levels=levels, n_nodes=n_nodes, pool=pool, predicted=True)

    @staticmethod
    def _caps_grow(caps, L: int, seen):
        """After an overflow in level L: twice the room from that level on, and at least what the device counted there (`seen`: the
        level's step blocks — only the steps up to the one that overflowed hold this attempt's totals, which max() keeps)."""
        levels = caps["levels"]
        while len(levels) <= L:          # (deeper than planned: one more level like the last)
            levels.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in levels[-1].items()})
        for li in range(L, len(levels)):
            lv = levels[li]
            for (st, c) in _CAP_COLS:
                lv[st][c] = 2 * int(lv[st][c]) + 8
            lv[0][14] = 2 * int(lv[0][14]) + 8
            for c in range(PREPARE_CLASSES):          # (a class that was empty may now be launched: with the class's upper bound)
                lv[4][16 + c] = PREPARE_LDS_BOUNDS[c]
        lv = levels[L]
        for (st, c) in _CAP_COLS:
            lv[st][c] = max(int(lv[st][c]), int(seen[st][c]))
        lv[0][14] = max(int(lv[0][14]), int(seen[0][15]))
        for li in range(L + 1, len(levels)):          # a frontier holds what the level before may create
            levels[li][0][14] = max(int(levels[li][0][14]), int(levels[li - 1][1][0] + levels[li - 1][5][2]))
        caps["n_nodes"] = 2 * int(caps["n_nodes"]) + sum(int(lv[0][14]) for lv in levels[L:])
        caps["pool"] = 2 * int(caps["pool"]) + sum(int(lv[5][1]) for lv in levels[L:])

    def _spec_enqueue(self, caps):
        """Every level of the forest enqueued from capacities (mprg_forest_level); no wait."""
        be = self.be
        n_words = DS_GLOBAL + (len(caps["levels"]) + 1 + SPEC_SPARE_LEVELS) * DS_LEVEL_WORDS
        d_ds = be.empty(8 * n_words)
        be.call("mprg_forest_state_init", be.ptr(d_ds), n_words, len(self.ok), be.stream)
        uoffs = np.zeros(MAX_CLUSTERS + 1, np.int32)
        for k_, o_ in self._uoff.items():
            uoffs[k_] = o_
        self._uoffs_host = uoffs
        small = bool(KM_MODE & 2)
        self._set(DS=d_ds, UNIFORMS=self._d_uni, LOOP_FORMS=(16 | 1) if (KM_MODE & 4) else ((1 | 8 | 2 | 4) if small else 1))
        self.F[FI["SIDE_STREAM"]] = be.side_ptr(0) if ((small or (KM_MODE & 4)) and KM_SIDE_STREAMS and be.n_side_streams >= 1 and be.side_ptr(0) != be.stream) else 0
        self.F[FI["UOFF_HOST"]] = uoffs.ctypes.data
        self.F[FI["CAP"] + CAP_BIG] = max(KM_BIG_BYTES - 1, 0)
        st = dict(caps=caps, d_ds=d_ds, n_words=n_words, reps=[], rec=[], done=0, tries=0)
        self._spec_levels(st, 0)
        return st

    def _spec_levels(self, st, L: int):
        """Node table and row pool for the whole forest (no growing on the way), then levels L .. of the capacities + the level that
        must find an empty frontier."""
        be, caps = self.be, st["caps"]
        C = FI["CAP"]
        need_nodes = max(self.cap_nodes, int(caps["n_nodes"]))
        if 8 * NODE_FIELDS * need_nodes > len(self.d_nodes):
            self.d_nodes = be.grown(self.d_nodes, 8 * NODE_FIELDS * self.n_nodes, 8 * NODE_FIELDS * need_nodes)
        self.cap_nodes = need_nodes
        pool_entries = max(int(caps["pool"]), 4)
        if 4 * pool_entries > self.pool_cap:
            self.d_pool = be.grown(self.d_pool, 4 * self.pool_used, 4 * pool_entries) if self.pool_used else be.empty(4 * pool_entries)
            self.pool_cap = 4 * pool_entries
        self._set(NODES=self.d_nodes, POOL=self.d_pool)
        self.F[C + CAP_NODES], self.F[C + CAP_POOL] = self.cap_nodes, self.pool_cap // 4
        levels = caps["levels"]
        del st["reps"][L:]
        for li in range(L, len(levels) + 1):
            st["reps"].append(self._level_speculative(li, levels[li] if li < len(levels) else None, int))
            self.counters["launches"] += 1

    def _spec_finish(self, st) -> bool:
        """The ONE wait of a forest enqueued from capacities, and one look at the device state.  A total that did not fit: the
        levels before it are kept, that level and the ones after it are enqueued again with more room (another wait) — or, when
        that has not settled it (SPEC_RETRIES) or the level holds a BIG clustering problem, left to the per-step host.  Returns
        whether the whole forest came from the device-counted path."""
        be = self.be
        while True:
            caps, d_ds = st["caps"], st["d_ds"]
            ds = be.download(d_ds, np.int64, st["n_words"])
            self.counters["syncs"] = self.counters.get("syncs", 0) + 1
            code = int(ds[DS_OVERFLOW])
            if not code and ds[DS_N]:          # (not reachable: the last level enqueued has no room and flags a frontier)
                code = 100 * (int(ds[DS_LEVEL]) + 1) + STEP_BEGIN
            n_done = (code // 100 - 1) if code else len(caps["levels"])
            self._spec_note_levels(st, ds, n_done)
            if not code:
                break
            L, step = code // 100 - 1, code % 100
            self.counters["plan_misses"] = self.counters.get("plan_misses", 0) + 1
            fb = ds[DS_GLOBAL + L * DS_LEVEL_WORDS:DS_GLOBAL + (L + 1) * DS_LEVEL_WORDS].reshape(6, HDR)
            if PLAN_TRACE:          # which totals did not fit (diagnostic)
                lv = caps["levels"][L] if L < len(caps["levels"]) else None
                over = [(st_, c_, int(fb[st_][c_]), int(lv[st_][c_])) for (st_, c_) in _CAP_COLS if lv is not None and fb[st_][c_] > lv[st_][c_]]
                import sys
                sys.stderr.write(f"[plan] overflow level {L} step {step} (try {st['tries'] + 1}): frontier {int(fb[0][15])} / {int(lv[0][14]) if lv is not None else 0}, "
                                 f"nodes {int(fb[0][16])} / {caps['n_nodes']}, pool {int(fb[0][17])} / {caps['pool']}; (step, column, seen, room): {over}\n")
            st["tries"] += 1
            if step == STEP_SIZES_SHAPE or st["tries"] > SPEC_RETRIES or L + 2 + SPEC_SPARE_LEVELS > (st["n_words"] - DS_GLOBAL) // DS_LEVEL_WORDS:
                self._forest_exact_resume(st, fb)
                return False
            self._caps_grow(caps, L, fb)
            be.call("mprg_forest_state_rewind", be.ptr(d_ds), st["n_words"], L, be.stream)
            self.n_nodes, self.pool_used = int(fb[0][16]), int(fb[0][17])          # (what a grown node table / row pool must keep)
            self._spec_levels(st, L)
        self.n_nodes, self.pool_used = int(ds[DS_NNODES]), int(ds[DS_POOL_USED])
        self._spec_failed = bool(ds[DS_NFAILED])
        self._set(N_NODES=self.n_nodes, POOL_USED=self.pool_used)
        self._plan = dict(key=self._roots_args[2], levels=st["rec"], n_nodes=self.n_nodes, pool_used=self.pool_used)
        return True

    def _spec_note_levels(self, st, ds, n_done: int):
        """Levels st['done'] .. n_done - 1 are complete on the device: their totals (the next plan), counters, leaf tables."""
        for li in range(st["done"], n_done):
            blk = ds[DS_GLOBAL + li * DS_LEVEL_WORDS:DS_GLOBAL + (li + 1) * DS_LEVEL_WORDS].reshape(6, HDR)
            st["rec"].append({s_: blk[s_].copy() for s_ in range(6)})
            st["rec"][-1]["rpc_idx"] = st["caps"]["levels"][li]["rpc_idx"]
            b0, b1, b2, b4 = blk[0], blk[1], blk[2], blk[4]
            self.counters["levels"] += 1
            self.counters["cells_all"] += float(b0[11])
            self.counters["cells_clustered"] += float(b2[3])
            self.counters["fits"] += int(b4[80])
            self.counters["kmeans_bytes"] += float(b4[85:86].view(np.float64)[0]) + float(b4[93:94].view(np.float64)[0])
            if b4[82]:
                raise MprgError("KMeans empty-cluster relocation: the selection ran out of frames (more than 5^10 samples in a fit)")
            if b4[14]:
                raise MprgError("k-mer dictionary: no hash seed separated the k-mers of a clustering problem (k-mer size > 16)")
            if b4[15]:
                raise MprgError("a k-mer count matrix has more than 4 194 304 features: beyond the KMeans kernels' pairwise-sum stack")
            self.levels.append(dict(f0=int(b0[13]), n=int(b0[14]), reps_pos=st["reps"][li][0], reps_len=st["reps"][li][1], reps_rows=int(b1[2])))
        st["done"] = max(st["done"], n_done)

    def _forest_exact_resume(self, st, fb):
        """The per-step host takes over at the level whose frontier block is fb (the levels before it came from the device-counted
        path and stand)."""
        self.counters["plan_resumes"] = self.counters.get("plan_resumes", 0) + 1
        f0, n = int(fb[0][13]), int(fb[0][15])
        self.n_nodes, self.pool_used = int(fb[0][16]), int(fb[0][17])
        self._set(N_NODES=self.n_nodes, POOL_USED=self.pool_used, NODES=self.d_nodes, POOL=self.d_pool)
        self._plan_rec = list(st["rec"])
        self._big_seen = False
        self._forest_exact_levels(f0, n)

    def _level_speculative(self, li: int, pl, cap):
        """Buffers of one level from the plan's totals `pl` (None: the level after the plan's last, which must find an empty
        frontier), then the whole level in one call.  Returns the level's (reps_pos, reps_len) buffers, which PRG assembly reads."""
        be = self.be
        F = self.F
        C = FI["CAP"]
        F[FI["LEVEL_INDEX"]] = li
        if pl is None:
            self._set(N=0, LVL=li)
            be.call("mprg_forest_level", F.ctypes.data, be.stream)
            return (None, None)
        h0, h1, h2, h3, h4, h5 = (pl[s_] for s_ in range(6))
        n = cap(h0[14])
        na, tcols, n_fused, n_other, n_gap = (cap(h0[q]) for q in (0, 1, 3, 4, 10))
        rpc_idx = pl["rpc_idx"]
        n_items = cap(h0[5 + rpc_idx])
        nsel, srows, ubytes, scols, n_dd = (cap(h1[q]) for q in (1, 2, 3, 4, 5))
        n_pq, n_wc, n_wr = (cap(h2[q]) for q in (0, 1, 2))
        P, table_bytes, flag_bytes, lo = (cap(h3[q]) for q in (0, 1, 2, 3))
        xd, wsd = cap(h4[0]), cap(h4[1])
        n_splits, n_child = cap(h5[0]), cap(h5[2])
        self._scratch(max(n, nsel, n_pq, P, 1))
        e, z = be.empty, be.zeros
        # the buffers that must arrive zeroed are places in ONE block: one fill per level instead of seven
        al = lambda x: (int(x) + 255) // 256 * 256
        zsizes = dict(MASK=4 * tcols, MAXRUN=4 * tcols, IVFLAG=8 * tcols, IVC=4)
        if P:
            zsizes.update(KM_STATUS=4 * P, ASSIGN=4 * lo, X=8 * xd)
        zoff, ztotal = {}, 0
        for name, nb in zsizes.items():
            zoff[name] = ztotal
            ztotal += al(max(nb, 16))
        zslab = z(ztotal)
        zbase = be.ptr(zslab)
        self._alive["ZSLAB"] = zslab
        bufs = dict(VIEWS=e(8 * VF * na), VIEW2NODE=e(8 * na), FUSED_LIST=e(4 * n_fused), OTHER_LIST=e(4 * n_other), MASK_WORK=e(12 * n_items),
                    GAP_WORK=e(8 * n_gap), VIEW_OUT=e(32 * na), IV_PACKED=e(12 * tcols), STACK=e(16 * tcols),
                    IV=e(12 * tcols), NIV=e(4 * na), STATUS=e(4 * na),
                    SUB=e(8 * VF * nsel), SELNODE=e(8 * nsel), DD_WORK=e(8 * n_dd))
        caps = {CAP_TCOLS: tcols, CAP_NFUSED: n_fused, CAP_NOTHER: n_other, CAP_ITEMS: n_items, CAP_NGAP: n_gap, CAP_SROWS: srows, CAP_UBYTES: ubytes,
                CAP_SCOLS: scols, CAP_NDD: n_dd, CAP_WC: n_wc, CAP_WR: n_wr, CAP_TABLE: table_bytes, CAP_FLAG: flag_bytes, CAP_LO: lo, CAP_XD: xd,
                CAP_WSD: wsd, CAP_NCHILD: n_child}
        for c_ in range(5):
            caps[CAP_CLS + c_] = cap(h4[2 + c_])
        for c_ in range(4):
            caps[CAP_LDS + c_] = int(h4[16 + c_])
        keep = (None, None)
        if nsel:
            R = max(srows, 1)
            bufs.update(UCODES=e(ubytes), GCODES=e(ubytes), HASHES=e(16 * R), ULEN=e(4 * R), REP_U=e(4 * R), REP_G=e(4 * R), D_OF_ROW=e(4 * R),
                        S_OF_ROW=e(4 * R), REPS_POS=e(4 * R), REPS_LEN=e(4 * R), SEQROW=e(4 * R), OCC_OFF=e(8 * (R + nsel)), SUMMARY=e(64 * nsel))
            keep = (bufs["REPS_POS"], bufs["REPS_LEN"])
        if n_pq:
            bufs.update(T1=e(8 * PF * n_pq), WORK_COLS=e(8 * n_wc), WORK_ROWS=e(8 * n_wr), CF_SCRATCH=e(12 * scols + 64), FURTHER=e(4 * n_pq))
        if P:
            bufs.update(PTAB0=e(8 * PF * P), TABLE=e(table_bytes), FLAG=e(flag_bytes), DV=e(4 * P), PTAB=e(8 * PF * P),
                        CLS_LISTS=e(4 * (PREPARE_CLASSES + 1) * P), NUM_CLUSTERS=e(4 * P), ACTIVE=e(4 * P), KM_INFO=e(64 * P),
                        WS=e(8 * wsd), LABELS=e(4 * lo))
        if n_splits:
            bufs.update(SPT=e(8 * PF * n_splits), SP=e(24 * n_splits), SPLITNODE=e(8 * n_splits), CHILD_SIZES=e(4 * n_child))
        for name in ("WORK_COLS", "WORK_ROWS") if not n_pq else ():
            F[FI[name]] = 0
        self._set(N=n, LVL=li, N_VIEWS=na, RPC_IDX=rpc_idx, NSEL=nsel, NPQ=n_pq, P=P, NSPLITS=n_splits, **bufs)
        for name, off in zoff.items():
            F[FI[name]] = zbase + off
        for q_, v_ in caps.items():
            F[C + q_] = v_
        be.call("mprg_forest_level", F.ctypes.data, be.stream)
        self._alive = {k: v for k, v in self._alive.items() if k in ("NODES", "META", "FAILED", "ERR_FIRST", "POOL", "ARENA", "HDR", "HDR_HOST", "DS", "UNIFORMS")}
        return keep

    # ------------------------------------------------------------------------------------------------ level
    def _forest_level(self, f0: int, n: int):
        be, L = self.be, self.L
        lvl = len(self.levels)
        # ---- S1: the frontier's views (nodes that are match intervals of their parent's scan need no kernel)
        self._scratch(n)
        self._set(F0=f0, N=n, LVL=lvl)
        h = self._step("frontier_count", n_hdr=13)
        rec = self._plan_note(None, 0, h)
        rec[0][13], rec[0][14] = f0, n
        na, total_cols, total_rows, n_fused, n_other = (int(x) for x in h[:5])
        items, n_gap, cells, cells_other = h[5:10], int(h[10]), float(h[11]), float(h[12])
        self.counters["cells_all"] += cells
        # the row chunk of a mask item is chosen so that a launch has >= ~1000 workgroups when the level offers that much work
        # (measured on MI355X: a 600 MB view streams at 4.9 TB/s with ~1200 workgroups of 512 rows x 1024 columns)
        rpc_idx = next((i for i in range(4) if items[i] >= 1024), 4)
        rec["rpc_idx"] = rpc_idx
        n_items = int(items[rpc_idx])
        d_views, d_v2n = be.empty(8 * VF * na), be.empty(8 * na)
        d_fl, d_ol, d_mw, d_gw = be.empty(4 * n_fused), be.empty(4 * n_other), be.empty(12 * n_items), be.empty(8 * n_gap)
        self._set(VIEWS=d_views, VIEW2NODE=d_v2n, FUSED_LIST=d_fl, OTHER_LIST=d_ol, MASK_WORK=d_mw, RPC_IDX=rpc_idx, GAP_WORK=d_gw,
                  N_VIEWS=na)
        self._step("frontier_fill")
        d_vout, d_ivp = be.empty(32 * na), be.empty(12 * total_cols)
        if na:
            d_mask = be.zeros(4 * total_cols)
            if n_other:
                be.call("mprg_column_masks", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(self.d_pool), be.ptr(d_mw), n_items,
                        1024 >> rpc_idx, be.ptr(d_mask), be.stream, work=cells_other)
            d_maxrun, d_stack, d_ivflag = be.zeros(4 * total_cols), be.empty(16 * total_cols), be.zeros(8 * total_cols)
            d_iv, d_niv, d_status, d_ivc = be.empty(12 * total_cols), be.empty(4 * na), be.empty(4 * na), be.zeros(4)
            lists = (be.ptr(d_fl), n_fused, be.ptr(d_ol), n_other) if n_fused else (None, 0, None, 0)
            be.call("mprg_partition", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(self.d_pool), na, be.ptr(d_mask), L,
                    be.ptr(d_gw), n_gap, be.ptr(d_maxrun), be.ptr(d_stack), be.ptr(d_ivflag), be.ptr(d_iv), be.ptr(d_niv),
                    be.ptr(d_status), be.ptr(d_vout), be.ptr(d_ivp), be.ptr(d_ivc), *lists, be.stream, work=cells)
            self.counters["launches"] += 2
        # ---- S2: leaf / multi-interval / clustering candidate; children of multi-interval nodes
        self._set(VIEW_OUT=d_vout, IV_PACKED=d_ivp)
        h = self._step("classify", n_hdr=7)
        self._plan_note(rec, 1, h)
        n_child_iv, nsel, tot_rows, tot_u, tot_cols, n_dd = (int(x) for x in h[:6])
        cells_sel = float(h[6])
        self._grow_nodes(self.n_nodes + n_child_iv)
        d_sub, d_selnode, d_ddw = be.empty(8 * VF * nsel), be.empty(8 * nsel), be.empty(8 * n_dd)
        self._set(N_NODES=self.n_nodes, SUB=d_sub, SELNODE=d_selnode, DD_WORK=d_ddw, NSEL=nsel)
        self._step("children")
        self.n_nodes += n_child_iv
        self.levels.append(dict(f0=f0, n=n, reps_pos=None, reps_len=None, reps_rows=0))
        n_child_cl = self._forest_cluster(lvl, d_sub, d_selnode, d_ddw, nsel, tot_rows, tot_u, tot_cols, n_dd, cells_sel) if nsel else 0
        self._set(N_NODES=self.n_nodes)
        return f0 + n, n_child_iv + n_child_cl

    # ------------------------------------------------------------------------------------------------ clustering
    def _forest_cluster(self, lvl, d_sub, d_selnode, d_ddw, nsel, tot_rows, tot_u, tot_cols, n_dd, cells_sel) -> int:
        """Clustering stage of a level (single non-match interval below a non-root node) + row groups of the leaves with
        several distinct rows.  Returns the number of cluster children appended to the node table."""
        be, K = self.be, self.L
        dd = self._dedupe(d_sub, self.d_pool, nsel, tot_rows, tot_u, work=2.0 * cells_sel, wr=(d_ddw, n_dd))
        # every selected view can end as a leaf whose alleles are its distinct rows (recursion_tree.py:272-274): the
        # device keeps the first-appearance lists of this level
        self.levels[lvl].update(reps_pos=dd["reps_pos"], reps_len=dd["reps_len"], reps_rows=tot_rows)
        # ---- S3: which candidates go on (recursion_tree.py:538-556 / :475-494, cluster_sequences.py:235-246)
        self._scratch(nsel)
        self._set(SUMMARY=dd["summary"])
        h = self._step("cluster_count", n_hdr=5)
        rec = self._plan_note(self._plan_rec[-1], 2, h)
        n_pq, n_wc, n_wr = (int(x) for x in h[:3])
        self.counters["cells_clustered"] += float(h[3])
        if n_pq == 0:
            return 0
        d_t1, d_wc, d_wr = be.empty(8 * PF * n_pq), be.empty(8 * n_wc), be.empty(8 * n_wr)
        d_scratch, d_further = be.empty(12 * tot_cols + 64), be.empty(4 * n_pq)
        self._set(T1=d_t1, WORK_COLS=d_wc, WORK_ROWS=d_wr, FURTHER=d_further, NPQ=n_pq)
        self._step("cluster_fill")
        # cluster_sequences.py:256: `while cluster_further(...)` is evaluated before any KMeans; a view whose rows are
        # already one-reference-like never uses its k-mer matrix, so the featurisation is only done for the others
        self._cluster_further(d_sub, d_t1, n_pq, 1, dd, None, None, d_wc, n_wc, d_wr, n_wr, d_scratch, d_further, None, None,
                              work=float(h[4]))
        # ---- S4: the clustering problems, their k-mer dictionaries
        self._scratch(n_pq)
        h = self._step("problems_count", n_hdr=4)
        self._plan_note(rec, 3, h)
        P, table_bytes, flag_bytes, lo = (int(x) for x in h)
        if P == 0:
            return 0
        d_ptab0, d_table, d_flag, d_V = be.empty(8 * PF * P), be.empty(table_bytes), be.empty(flag_bytes), be.empty(4 * P)
        self._set(PTAB0=d_ptab0, DV=d_V, P=P)
        self._step("problems_fill")
        if KD_PARTS_FROM and K <= 16 and flag_bytes >= KD_PARTS_FROM * P:
            # problems of (on average) this many k-mer occurrences: many workgroups per problem (the top of one deep alignment holds
            # 2 x 10^8 occurrences in ONE problem)
            d_pc = be.empty(4 * P * KD_PARTS)
            be.call("mprg_kmer_dictionary_parts", be.ptr(d_sub), be.ptr(d_ptab0), P, K, be.ptr(dd["ucodes"]), be.ptr(dd["ulen"]),
                    be.ptr(dd["seqrow"]), be.ptr(dd["occ_off"]), be.ptr(d_table), be.ptr(d_flag), be.ptr(d_V), KD_PARTS, be.ptr(d_pc), be.stream)
        else:
            be.call("mprg_kmer_dictionary", be.ptr(d_sub), be.ptr(d_ptab0), P, K, be.ptr(dd["ucodes"]), be.ptr(dd["ulen"]),
                    be.ptr(dd["seqrow"]), be.ptr(dd["occ_off"]), be.ptr(d_table), be.ptr(d_flag), be.ptr(d_V), be.stream)
        # ---- S5: count matrices, workspaces, launch classes, biggest fits first
        self._scratch(P)
        h = self._step("sizes_count", n_hdr=21)
        self._plan_note(rec, 4, h)
        if h[14]:
            raise MprgError("k-mer dictionary: no hash seed separated the k-mers of a clustering problem (k-mer size > 16)")
        if h[15]:
            raise MprgError("a k-mer count matrix has more than 4 194 304 features: beyond the KMeans kernels' pairwise-sum stack")
        # a level of a FEW BIG problems (the top of one deep alignment): every round's general-form fit at once (_kloop_rounds,
        # mprg_kmeans_speculative_kinfo) — their workspaces hold the restart slots of nine rounds
        spec_k = bool(KM_BIG_BYTES) and int(h[16:17 + PREPARE_CLASSES].max()) >= KM_BIG_BYTES and 0 < P <= KM_SPEC_PROBLEMS and N_INIT * (MAX_CLUSTERS - 1) <= 16 * 9
        if spec_k:
            self.F[FI["N_INIT"]] = N_INIT * (MAX_CLUSTERS - 1)
            h = self._step("sizes_count", n_hdr=21)
            self.F[FI["N_INIT"]] = N_INIT
            self._plan_note(rec, 4, h)
        x_doubles, ws_doubles, n_wc, n_wr = int(h[0]), int(h[1]), int(h[7]), int(h[8])
        d_ptab, d_cls = be.empty(8 * PF * P), be.empty(4 * (PREPARE_CLASSES + 1) * P)
        d_numcl, d_active, d_kinfo = be.empty(4 * P), be.empty(4 * P), be.empty(20 * P)
        d_info, d_st, d_further = be.empty(64 * P), be.zeros(4 * P), be.zeros(4 * P)
        d_wc, d_wr = be.empty(8 * n_wc), be.empty(8 * n_wr)
        d_x, d_ws = be.zeros(8 * x_doubles), be.empty(8 * ws_doubles)
        d_labels, d_assign = be.empty(4 * lo), be.zeros(4 * lo)
        self._set(PTAB=d_ptab, CLS_LISTS=d_cls, NUM_CLUSTERS=d_numcl, ACTIVE=d_active, KINFO=d_kinfo, KM_INFO=d_info, KM_STATUS=d_st,
                  FURTHER=d_further, WORK_COLS=d_wc, WORK_ROWS=d_wr)
        self._step("sizes_fill")
        big_level = bool(KM_BIG_BYTES) and int(h[16:17 + PREPARE_CLASSES].max()) >= KM_BIG_BYTES          # (the level's largest count matrix + means)
        if big_level:          # (a big problem's occurrences — 10^8 at the top of one deep alignment — over many workgroups)
            be.call("mprg_kmer_counts_parts", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(dd["ucodes"]), be.ptr(dd["ulen"]),
                    be.ptr(dd["seqrow"]), be.ptr(dd["occ_off"]), be.ptr(d_table), be.ptr(d_x), 128, be.stream)
        else:
            be.call("mprg_kmer_counts", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(dd["ucodes"]), be.ptr(dd["ulen"]),
                    be.ptr(dd["seqrow"]), be.ptr(dd["occ_off"]), be.ptr(d_table), be.ptr(d_x), be.stream)
        # mprg_kmeans_prepare, one launch per class of LDS need (every workgroup of a launch allocates the launch's lds_bytes)
        prep_work = 8.0 * x_doubles
        d_xb = None
        for c in range(PREPARE_CLASSES + 1):
            n_c = int(h[2 + c])
            if not n_c:
                continue
            lst = be.ptr(d_cls) + 4 * c * P
            if c < PREPARE_CLASSES:
                be.call("mprg_kmeans_prepare", be.ptr(d_ptab), n_c, be.ptr(d_x), be.ptr(d_ws), lst, n_c, int(h[16 + c]), 0, 0, be.stream,
                        work=prep_work)
            elif big_level:          # (their fits take mprg_kmeans_fit_wide below: the byte matrix; no seeding tables beyond KM_NO_TABLES_BYTES)
                d_xb = be.empty(8 * x_doubles)
                be.call("mprg_kmeans_prepare_big", be.ptr(d_ptab), be.ptr(d_x), be.ptr(d_ws), lst, n_c, be.ptr(d_xb),
                        0 if int(h[16 + PREPARE_CLASSES]) >= KM_NO_TABLES_BYTES else 1, be.stream, work=prep_work)
            else:
                be.call("mprg_kmeans_prepare", be.ptr(d_ptab), n_c, be.ptr(d_x), be.ptr(d_ws), 0, 0, 0, lst, n_c, be.stream, work=prep_work)
            prep_work = 0.0
        self.counters["launches"] += 3
        # ---- S6: cluster_sequences.py:256-274, the whole `while cluster_further: k += 1; KMeans(k)` loop of every problem inside the
        #      problem's own workgroup (mprg_cluster_loop): one launch per workgroup form for the level, no control step, header or
        #      host wait per round.  The general form and the small forms are independent launches: side by side on a side stream.
        #      MPRG_KLOOP=rounds keeps the per-round launches of rounds 1-3 (k_kl_advance + fit lists + mprg_cluster_further).
        km_events, cf_events = [], []
        big = big_level
        self.counters["max_problem_bytes"] = max(int(self.counters.get("max_problem_bytes", 0)), int(h[16:17 + PREPARE_CLASSES].max()))
        if big:
            self._big_seen = True
        fused = self.kloop_fused and not big
        if fused:
            uoffs = np.zeros(MAX_CLUSTERS + 1, np.int32)
            for k_, o_ in self._uoff.items():
                uoffs[k_] = o_
            small = bool(KM_MODE & 2)
            loop_args = (be.ptr(d_sub), be.ptr(d_ptab), P, N_INIT, be.ptr(self._d_uni), uoffs.ctypes.data, be.ptr(d_x), be.ptr(d_ws),
                         be.ptr(dd["d_of_row"]), be.ptr(dd["gcodes"]), be.ptr(d_scratch), be.ptr(d_labels), be.ptr(d_assign), be.ptr(d_info),
                         be.ptr(d_st), be.ptr(d_numcl), be.ptr(d_active), be.ptr(self.d_hdr))
            lds = bool(KM_MODE & 4)
            side = (small or lds) and KM_SIDE_STREAMS and be.profile is None and be.n_side_streams >= 1
            if side:
                be.fork(1)
            if lds:          # the LDS form, a launch per class (beside them, given a side stream, the general form for the problems no class
                #              holds); then the general form for the rounds that left the classes on the way
                if side:
                    be.call("mprg_cluster_loop", *loop_args, 1 | 32, be.stream, label=LOOP_GENERAL)
                be.call("mprg_cluster_loop", *loop_args, 16, be.side_ptr(0) if side else be.stream, side=0 if side else None, label=LOOP_SMALL)
                ev = self._last_event(LOOP_SMALL)
                km_events.append(ev and ev + (LOOP_SMALL,))
                if side:
                    be.join(1)
                    side = False
            be.call("mprg_cluster_loop", *loop_args, 1 | (8 if (small and not lds) else 0), be.stream, label=LOOP_GENERAL)
            ev = self._last_event(LOOP_GENERAL)
            km_events.append(ev and ev + (LOOP_GENERAL,))
            if small and not lds:
                be.call("mprg_cluster_loop", *loop_args, 2 | 4, be.side_ptr(0) if side else be.stream, side=0 if side else None, label=LOOP_SMALL)
                ev = self._last_event(LOOP_SMALL)
                km_events.append(ev and ev + (LOOP_SMALL,))
            if side:
                be.join(1)
            self.counters["launches"] += 5 if lds else (2 if small else 1)
        else:
            self._kloop_rounds(P, d_sub, d_ptab, d_kinfo, d_x, d_ws, d_labels, d_assign, d_info, d_st, d_wc, n_wc, d_wr, n_wr, d_scratch,
                               d_further, dd, km_events, cf_events, wide=big, d_xb=d_xb, spec_k=spec_k and big, lo=lo)
        # ---- S7: MultiClusterNodes and their children (cluster_sequences.py:276-296, recursion_tree.py:457-469)
        self._scratch(P)
        h = self._step("splits_count", n_hdr=HDR)
        self._plan_note(rec, 5, h[:3])
        n_splits, rows_sp, n_child = (int(x) for x in h[:3])
        fits, cf_cells = int(h[80]), float(h[84:85].view(np.float64)[0])
        kb = {KM_LDS_ENTRY: float(h[81:82].view(np.float64)[0]), "mprg_kmeans_fit": float(h[85:86].view(np.float64)[0]),
              "mprg_kmeans_fit_small": float(h[93:94].view(np.float64)[0])}
        km_bytes = sum(kb.values())
        if fused:          # a fused launch's algorithmic bytes: its fits' 8 D V (iterations + n_init) + the cells its cluster_further visits
            kb = {LOOP_GENERAL: kb["mprg_kmeans_fit"], LOOP_SMALL: kb["mprg_kmeans_fit_small"]}
            kb[LOOP_SMALL if (KM_MODE & 6) else LOOP_GENERAL] += cf_cells
        if h[82]:
            raise MprgError("KMeans empty-cluster relocation: the selection ran out of frames (more than 5^10 samples in a fit)")
        self.counters["fits"] += fits
        self.counters["kmeans_bytes"] += km_bytes
        # algorithmic bytes are known only after the fits: 8 D V (iterations + n_init)
        for entry, nbytes in kb.items():
            self._credit([e[:2] for e in km_events if e and e[2] == entry], nbytes)
        self._credit(cf_events, cf_cells)
        if n_splits == 0:
            return 0
        self._pool_reserve(rows_sp)
        d_spt, d_sp, d_splitnode, d_sizes = be.empty(8 * PF * n_splits), be.empty(24 * n_splits), be.empty(8 * n_splits), be.empty(4 * n_child)
        self._set(SPT=d_spt, SP=d_sp, SPLITNODE=d_splitnode, CHILD_SIZES=d_sizes, NSPLITS=n_splits, POOL_USED=self.pool_used)
        self._step("splits_fill")
        # rowidx (parents' lists) and pool_out (children's lists) are the same pool, disjoint regions
        be.call("mprg_split_children", be.ptr(d_sub), be.ptr(self.d_pool), be.ptr(d_spt), n_splits, be.ptr(d_sp),
                be.ptr(dd["d_of_row"]), be.ptr(dd["s_of_row"]), be.ptr(d_assign), be.ptr(self.d_pool), be.ptr(d_sizes), be.stream)
        self.counters["launches"] += 1
        self._grow_nodes(self.n_nodes + n_child)
        self._set(N_NODES=self.n_nodes)
        self._step("split_children")
        self.pool_used += rows_sp
        self.n_nodes += n_child
        self._set(POOL_USED=self.pool_used)
        return n_child


    def _kloop_rounds(self, P, d_sub, d_ptab, d_kinfo, d_x, d_ws, d_labels, d_assign, d_info, d_st, d_wc, n_wc, d_wr, n_wr, d_scratch,
                      d_further, dd, km_events, cf_events, wide=False, d_xb=None, spec_k=False, lo=0):
        """The clustering loop as one set of launches per round k (rounds 1-3's shape; MPRG_KLOOP=rounds): the control step settles
        the previous round on the device (k_kl_advance), a retired problem's workgroups return at once.
        spec_k (a level of a few BIG problems): the general-form fit of EVERY round goes out first, in one launch of wide workgroups —
        ninety per problem instead of ten nine times over; round k's other launches, its mprg_cluster_further and the control step of
        round k + 1 then work on slice k - 2 of the labels / km_info / km_status arrays, where the round's fits already are."""
        be = self.be
        d_fl = be.empty(4 * len(KM_LISTS) * P)
        self._set(FIT_LISTS=d_fl, KM_MODE=KM_MODE, WS=d_ws)          # (WS: the control step asks a problem's workspace whether K6 made its tables)
        fit_args = (be.ptr(self._d_uni), be.ptr(d_x), be.ptr(d_ws))
        n_k = MAX_CLUSTERS - 1
        if spec_k:
            uoffs = np.zeros(MAX_CLUSTERS + 1, np.int32)
            for k_, o_ in self._uoff.items():
                uoffs[k_] = o_
            d_labels, d_info, d_st = be.empty(4 * n_k * max(lo, 1)), be.empty(64 * n_k * P), be.zeros(4 * n_k * P)
            d_ki = be.empty(20 * n_k * P)
            be.call("mprg_kmeans_speculative_kinfo", be.ptr(d_ptab), P, N_INIT, KM_MODE, uoffs.ctypes.data, lo, be.ptr(d_ki), be.ptr(d_ws), be.stream)
            be.call("mprg_kmeans_fit_wide", be.ptr(d_ptab), be.ptr(d_ki), 0, n_k * P, N_INIT, *fit_args, be.ptr(d_labels), be.ptr(d_info), be.ptr(d_st),
                    be.ptr(d_xb) if d_xb is not None else 0, be.stream)
            ev = self._last_event("mprg_kmeans_fit_wide")
            km_events.append(ev and ev + ("mprg_kmeans_fit",))
            self.counters["launches"] += 3
            self.counters["speculative_levels"] = self.counters.get("speculative_levels", 0) + 1
        # the arrays of round k (spec_k: slice k - 2; else the level's)
        lab_of = lambda k: be.ptr(d_labels) + (4 * (k - 2) * lo if spec_k else 0)
        info_of = lambda k: be.ptr(d_info) + (64 * (k - 2) * P if spec_k else 0)
        st_of = lambda k: be.ptr(d_st) + (4 * (k - 2) * P if spec_k else 0)
        for k in range(2, MAX_CLUSTERS + 2):
            if spec_k:          # the control step settles round k - 1: that round's results
                self.F[FI["KM_INFO"]], self.F[FI["KM_STATUS"]] = info_of(max(k - 1, 2)), st_of(max(k - 1, 2))
            hk = self._step("kloop_advance", k, n_hdr=HDR)
            if k > MAX_CLUSTERS or hk[83] == 0:
                break
            # the round's fits, already sorted into launch lists by the control step; the lists are independent launches of
            # different kernels: side by side on side streams, so that one list's tail (its last, longest fits) overlaps the others
            todo = [(c, int(hk[86 + c])) for c in range(len(KM_LISTS)) if hk[86 + c] and not (spec_k and KM_LISTS[c][1] is None)]
            # (not while per-entry-point events are recorded: they would time launches that overlap)
            n_side = len(todo) if (len(todo) > 1 and KM_SIDE_STREAMS and be.profile is None and be.n_side_streams >= len(todo)) else 0
            if n_side:
                be.fork(n_side)
            for q, (c, n_c) in enumerate(todo):
                entry, cls = KM_LISTS[c]
                list_entry = entry                    # (the list's algorithmic bytes are credited under this name, split or not)
                lst = be.ptr(d_fl) + 4 * c * P
                stream = be.side_ptr(q) if n_side else be.stream
                outs = (lab_of(k), info_of(k), st_of(k), stream)
                if (wide and cls is None) or n_c <= KM_SPLIT_BELOW:          # (wide: a level with BIG problems, its general-form fits)
                    if wide and cls is None:          # (d_xb: the byte matrix mprg_kmeans_prepare_big wrote for the level's big problems)
                        entry = "mprg_kmeans_fit_wide"
                        be.call(entry, be.ptr(d_ptab), be.ptr(d_kinfo), lst, n_c, N_INIT, *fit_args, *outs[:-1], be.ptr(d_xb) if d_xb is not None else 0,
                                outs[-1], side=q if n_side else None)
                    else:
                        entry = "mprg_kmeans_fit_split"
                        be.call(entry, be.ptr(d_ptab), be.ptr(d_kinfo), lst, n_c, N_INIT, *fit_args, *outs, side=q if n_side else None)
                elif cls is None:
                    be.call(entry, be.ptr(d_ptab), be.ptr(d_kinfo), lst, n_c, N_INIT, *fit_args, 0, 0, 0, 0, *outs, side=q if n_side else None)
                else:
                    be.call(entry, be.ptr(d_ptab), be.ptr(d_kinfo), lst, n_c, cls, N_INIT, *fit_args, *outs, side=q if n_side else None)
                ev = self._last_event(entry)
                km_events.append(ev and ev + (list_entry,))
                self.counters["launches"] += 1 + (entry in ("mprg_kmeans_fit_split", "mprg_kmeans_fit_wide"))
            if n_side:
                be.join(n_side)
            self._cluster_further(d_sub, d_ptab, P, k, dd, lab_of(k), d_assign, d_wc, n_wc, d_wr, n_wr, d_scratch, d_further, info_of(k), d_kinfo)
            cf_events.append(self._last_event("mprg_cluster_further"))
        if spec_k:
            self._alive["SPEC_K"] = (d_labels, d_info, d_st, d_ki)          # (until the level ends: the stream still reads them)
            if be.profile is not None:
                # (profiled passes only) the algorithmic bytes of EVERYTHING the one launch fitted — the rounds the loop reached are what
                # `kmeans_bytes` counts (k_kl_advance), the others are work the reference never does: reported beside it, not inside it
                ki = be.download(d_ki, np.int32, 5 * n_k * P).reshape(n_k * P, 5)
                info = be.download(d_info, np.float64, 8 * n_k * P).reshape(n_k * P, 8)
                pt = be.download(d_ptab, np.int64, PF * P).reshape(P, PF)
                ran = ki[:, 1] > 0
                dv = (pt[ki[:, 0], 1] * pt[ki[:, 0], 7]).astype(np.float64)
                self.counters["kmeans_bytes_launched_at_once"] = self.counters.get("kmeans_bytes_launched_at_once", 0.0) + \
                    float((8.0 * dv[ran] * (info[ran, 4] + N_INIT)).sum())

    def _last_event(self, name):
        prof = self.be.profile
        return (name, len(prof[name]) - 1) if prof is not None and prof.get(name) else None

    def _credit(self, events, total_bytes: float):
        """Roofline accounting of launches whose algorithmic bytes the device only knows afterwards: the level's total goes
        to the level's last launch of that entry point (sums per entry point are what the reports use)."""
        events = [e for e in events if e is not None]
        if events:
            name, i = events[-1]
            a0, a1, _ = self.be.profile[name][i]
            self.be.profile[name][i] = (a0, a1, float(total_bytes))

    def _cluster_further(self, d_sub, d_prob, n_probs, k, dd, d_labels, d_assign, d_wc, n_wc, d_wr, n_wr, d_scratch, d_further, d_info,
                         d_kinfo, work=0.0):
        be = self.be
        p = lambda b: (b if isinstance(b, int) else be.ptr(b)) if b is not None else None          # (a buffer, or an address inside one)
        # (no view has more rows than the forest's largest root: the launch that only tall problems need is left out below that)
        be.call("mprg_cluster_further_bounded", be.ptr(self.d_arena), be.ptr(d_sub), be.ptr(self.d_pool), be.ptr(d_prob), n_probs, k,
                be.ptr(dd["d_of_row"]), p(d_labels), p(d_assign), be.ptr(d_wc), n_wc, be.ptr(d_wr), n_wr, be.ptr(d_scratch),
                be.ptr(d_further), p(d_info), be.ptr(dd["gcodes"]), p(d_kinfo), int(self.F[FI["MAX_ROWS"]]), be.stream, work=work,
                label="mprg_cluster_further")
        self.counters["launches"] += 2

    # ------------------------------------------------------------------------------------------------ host views of the tables
    @property
    def tab(self) -> Dict[str, np.ndarray]:
        """The node table as host arrays (downloaded on first use: tree dumps, the PRG index, object materialisation)."""
        if self._tab is None:
            t = self.be.download(self.d_nodes, np.int64, NODE_FIELDS * self.n_nodes).reshape(self.n_nodes, NODE_FIELDS)
            self._tab = dict(msa=t[:, N_MSA], parent=t[:, N_PARENT], level=t[:, N_LEVEL], rows_off=t[:, N_ROWS_OFF], nrows=t[:, N_NROWS],
                             col0=t[:, N_COL0], ncols=t[:, N_NCOLS], kind=t[:, N_KIND], first_child=t[:, N_FIRST_CHILD],
                             n_child=t[:, N_NCHILD], lvl=t[:, N_LVL], reps_off=t[:, N_REPS_OFF], nseq=t[:, N_NSEQ],
                             allele_chars=t[:, N_ACHARS], special=(t[:, N_FLAGS] & NF_SPECIAL) != 0)
        return self._tab

    def node_rows(self, ni: int, pool: np.ndarray) -> Optional[np.ndarray]:
        """MSA rows of node ni (None: all rows)."""
        t = self.tab
        ro = int(t["rows_off"][ni])
        return None if ro < 0 else pool[ro:ro + int(t["nrows"][ni])]


# ======================================================================================================= PRG assembly
def _special_leaf_alleles(self: "ForestEngine", rows: np.ndarray) -> Dict[int, List[str]]:
    """Leaves whose columns contain ambiguity codes or N: IUPAC expansion on the host (utils/seq_utils.py:116-153).
    Rare; fetches the leaf's distinct rows from the device lists.  rows: {node, its node-table row} per such leaf."""
    pool = self.pool_host() if self.pool_used else np.zeros(0, np.int64)
    out: Dict[int, List[str]] = {}
    cache: Dict[int, np.ndarray] = {}
    for r in rows[np.argsort(rows[:, 0])]:
        lf, nd = int(r[0]), r[1:]
        lv = int(nd[N_LVL])
        if lv not in cache:
            cache[lv] = self.be.download(self.levels[lv]["reps_pos"], np.int32, self.levels[lv]["reps_rows"]).astype(np.int64)
        ro = int(nd[N_REPS_OFF])
        rp = cache[lv][ro:ro + int(nd[N_NSEQ])]
        rows_off = int(nd[N_ROWS_OFF])
        mrows = rp if rows_off < 0 else pool[rows_off + rp]
        mi = int(nd[N_MSA])
        codes = self.codes[mi]
        block = codes[mrows, int(nd[N_COL0]):int(nd[N_COL0]) + int(nd[N_NCOLS])]
        seqs = [_ACGT[x[x != CODE_GAP]].tobytes().decode() for x in block]
        try:
            out[lf] = expand_sequences(seqs)
        except SequenceCurationError as err:
            self.failed[mi] = True
            self.errors[mi] = err
            out[lf] = ["A"]
    return out


def assemble_prgs(self: ForestEngine, want_index: bool = False, as_bytes: bool = False, lazy: bool = False, export: bool = False,
                  ring: int = 0, pack_alignments: bool = False):
    """PRG string of every alignment of the batch (None for loci dropped by the curation policy).  lazy: returns a function
    that waits for the text's copy to the host and returns the list — the copy then overlaps whatever the caller enqueues next
    (as_bytes views stay valid until the second following assemble_prgs of this engine's backend).
    want_index: self.prg_index_entries(i) afterwards.  export: self.exported = the trees as per-locus slices of three arrays
    (records, rows, PRG index: mprg_forest_export_* in include/mprg.h) for the update data structure.
    pack_alignments (with export): self.exported also holds every alignment at four bits per cell (mprg_export_alignments).
    ring: which set of pinned buffers the copies cycle through (backend.download_async groups 4 * ring ..): a caller that builds
    a side batch on a backend whose main ring is in use by a pipeline (pipeline.py: the object path of a chunk) takes its own.
    Device (mprg_forest_assemble_*): preorder ranks and site numbers, text lengths bottom-up, text offsets top-down over the
    node table; every leaf's alleles (mprg_emit_alleles) and every marker.  Host: the rare leaves with ambiguity codes.
    reference: PrgBuilder.build_prg prg_builder.py:100-105; traversals recursion_tree.py:194-201, :222-239, :266-300."""
    be = self.be
    n, M = self.n_nodes, len(self._msas)
    self._index = None
    self.exported = None
    if n == 0:
        self._asm, self._site_count = np.zeros((0, ASM_FIELDS), np.int64), np.zeros(M, np.int64)
        self._index, self._host_index = (np.zeros((0, 3), np.int32), np.zeros(M + 1, np.int64)), {}
        if export:
            self.exported = dict(records=np.zeros((0, 8), np.int32), rows=np.zeros(0, np.int32), node_bounds=np.zeros(M + 1, np.int64),
                                 row_bounds=np.zeros(M + 1, np.int64), index=self._index[0], index_bounds=self._index[1])
        nothing = lambda: [None] * M
        nothing.buffer, nothing.base, nothing.length = np.zeros(0, np.uint8), np.zeros(M, np.int64), np.full(M, -1, np.int64)
        return nothing if lazy else nothing()
    lv_arr = np.zeros((len(self.levels), 4), np.int64)
    for i, lv in enumerate(self.levels):
        lv_arr[i] = (lv["f0"], lv["n"], be.ptr(lv["reps_pos"]) if lv["reps_pos"] is not None else 0,
                     be.ptr(lv["reps_len"]) if lv["reps_len"] is not None else 0)
    d_root = be.upload(self.root_of)
    cap = 1024
    host_leaf: Dict[int, List[str]] = {}
    while True:          # leaves with ambiguity codes / N in their columns: host expansion
        d_list = be.empty(8 * (NODE_FIELDS + 1) * cap)
        self._set(NODES=self.d_nodes, N_NODES=n, FAILED=self.d_failed, SPECIAL_LIST=d_list, SPECIAL_CAP=cap, ROOT_OF=d_root, POOL=self.d_pool)
        n_sp = int(self._step("assemble_special", n_hdr=1)[0])
        if n_sp <= cap:
            break
        cap = n_sp
    d_patch, n_patch = None, 0
    if n_sp:
        rows = be.download(d_list, np.int64, (NODE_FIELDS + 1) * n_sp).reshape(n_sp, NODE_FIELDS + 1)
        before = self.failed.copy()
        host_leaf = _special_leaf_alleles(self, rows)
        patch = np.asarray([[lf, len(q), sum(len(s) for s in q)] for lf, q in host_leaf.items()], np.int64)
        d_patch, n_patch = be.upload(patch), len(patch)
        if (self.failed != before).any():
            self.d_failed = be.upload(self.failed.astype(np.int32))
            self._set(FAILED=self.d_failed)
    d_asm = be.empty(8 * ASM_FIELDS * n)
    d_vm, d_vn, d_vp = be.empty(8 * M), be.empty(8 * n), be.empty(8 * n)          # one-column tables
    d_nsites, d_mbase = be.empty(8 * M), be.empty(8 * 4 * M)
    self._set(ASM=d_asm, PATCH=d_patch, N_PATCH=n_patch, N_LEVELS=len(self.levels), VALS_MSA=d_vm, VALS_NODE=d_vn, VALS_POS=d_vp,
              N_SITES=d_nsites, MSA_BASE=d_mbase, SCAN_TMP=be.empty(8 * VC * (max(n, M) // 2048 + 2)))
    self.F[FI["LEVELS"]] = lv_arr.ctypes.data
    h = self._step("assemble_layout", n_hdr=2)
    total_chars, n_jobs = int(h[0]), int(h[1])
    d_out, d_jobs = be.empty(total_chars), be.empty(32 * max(n_jobs, 1))
    d_index = be.empty(12 * max(n_jobs, 1)) if (want_index or export) else None
    self._set(OUT=d_out, JOBS=d_jobs, INDEX_OUT=d_index)
    self._step("assemble_emit")
    if n_jobs:
        be.call("mprg_emit_alleles", be.ptr(self.d_arena), be.ptr(d_jobs), n_jobs, be.ptr(d_out), be.stream,
                work=float(2 * total_chars))
        self.counters["launches"] += 1
    d_rec = d_rows = d_aln = None
    n_ex_rows = aln_bytes = 0
    if export:
        n_ex_rows = int(self._step("export_count", n_hdr=1)[0])
        d_rec, d_rows = be.empty(32 * n), be.empty(4 * max(n_ex_rows, 1))
        self._set(EX_RECORDS=d_rec, EX_ROWS=d_rows)
        self._step("export_fill")
        if pack_alignments:          # the loci's alignments at four bits per cell (mprg_export_alignments), for the update_DS members
            S_, C_ = self.meta_arr[:, 4], self.meta_arr[:, 5]
            row_base = np.concatenate([[0], np.cumsum(S_)]).astype(np.int64)
            sizes = S_ * ((C_ + 1) // 2)
            aln_off = (np.cumsum(sizes) - sizes).astype(np.int64)
            aln_bytes = int(sizes.sum())
            d_aln = be.empty(max(aln_bytes, 16))
            d_rb, d_ao = be.upload(row_base), be.upload(aln_off)
            be.call("mprg_export_alignments", be.ptr(self.d_arena), be.ptr(self.d_meta), be.ptr(d_rb), be.ptr(d_ao), M, int(row_base[-1]),
                    be.ptr(d_aln), be.stream, work=1.5 * float((S_ * C_).sum()))
            self.counters["launches"] += 1
    # small copies before the big asynchronous one (a copy queued behind 2.5 GB on the DMA engine waits for it).
    # per alignment: start of its PRG in the batch text, first allele job / index entry, first node of its preorder run,
    # first entry of its exported rows
    mb = be.download(d_mbase, np.int64, 4 * M).reshape(4, M).T          # (column-major on the device)
    msa_base = mb[:, 0].copy()
    msa_len = np.diff(np.concatenate([msa_base, [total_chars]]))

    def bounds(col, total):
        """[first, end) per alignment of a per-locus contiguous layout (alignments without a tree: empty)."""
        first = col.copy()
        nxt = total
        for m in range(M - 1, -1, -1):               # (M is the batch's alignments: a short loop next to the work above)
            if first[m] < 0:
                first[m] = nxt
            nxt = first[m]
        return np.concatenate([first, [total]])

    self._asm = self._site_count = None
    self._d_asm, self._d_nsites = d_asm, d_nsites
    node_bounds = np.concatenate([mb[:, 2], [n]])
    waits = []
    if want_index or export:
        ix_host, w_ = be.download_async(d_index, 12 * n_jobs, group=4 * ring + 1)
        waits.append(w_)
        job_bounds = bounds(np.where(self.root_of >= 0, mb[:, 1], -1), n_jobs)
        self._index = (ix_host.view(np.int32).reshape(-1, 3), job_bounds)
    if export:
        rec_host, w1 = be.download_async(d_rec, 32 * n, group=4 * ring + 2)
        rows_host, w2 = be.download_async(d_rows, 4 * n_ex_rows, group=4 * ring + 3)
        waits += [w1, w2]
        self.exported = dict(records=rec_host.view(np.int32).reshape(-1, 8), rows=rows_host.view(np.int32), node_bounds=node_bounds,
                             row_bounds=bounds(np.where(self.root_of >= 0, mb[:, 3], -1), n_ex_rows),
                             index=self._index[0], index_bounds=self._index[1])
        if d_aln is not None:
            aln_host, w3 = be.download_async(d_aln, aln_bytes, group=8 + ring)
            waits.append(w3)
            self.exported.update(alignments=aln_host, alignment_off=aln_off, alignment_bytes=sizes.astype(np.int64))
    if host_leaf:
        _ = self.asm, self.tab
    buf, wait = be.download_async(d_out, total_chars, group=4 * ring)
    waits.append(wait)
    # the text of the host-expanded leaves: placed now (this engine's tables may belong to the next batch by the time the
    # caller collects the text), written into the buffer once the copy has landed
    text_patches, self._host_index = [], {}
    if host_leaf:
        A, msa = self.asm, self.tab["msa"]
        for lf, seqs in host_leaf.items():
            mi = int(msa[lf])
            if self.failed[mi]:
                continue
            pos, site = int(A[lf, A_START]), int(A[lf, A_SITE])
            base = int(msa_base[mi])
            many = len(seqs) > 1
            parts = [f" {site} "] if many else []
            at = pos + (len(parts[0]) if many else 0)
            for i, q in enumerate(seqs):
                parts.append(q)
                self._host_index.setdefault(mi, []).append([at - base, at - base + len(q), int(A[lf, A_PRE])])
                at += len(q)
                if many:
                    parts.append(f" {site + 1 if i < len(seqs) - 1 else site} ")
                    at += len(parts[-1])
            text_patches.append((pos, "".join(parts).encode()))
    spans = list(zip(msa_base.tolist(), msa_len.tolist(), self.failed.tolist()))

    def finish():
        """Wait for the copies; one bytes-like (or str) per alignment."""
        for w_ in waits:
            w_()
        for pos, txt in text_patches:
            buf[pos:pos + len(txt)] = np.frombuffer(txt, np.uint8)
        if as_bytes:          # zero-copy views into the batch buffer (ASCII); valid until the buffer's slot is reused
            mv = memoryview(buf)
            return [None if bad else mv[a:a + ln] for a, ln, bad in spans]
        whole = buf.tobytes()
        return [None if bad else whole[a:a + ln].decode() for a, ln, bad in spans]

    # (for callers that hand the text to native code: the pinned buffer and every alignment's span in it; valid after finish())
    finish.buffer, finish.base, finish.length = buf, msa_base, np.where(self.failed, -1, msa_len)
    return finish if lazy else finish()


def _asm_host(self: ForestEngine) -> np.ndarray:
    """The assembly table (node ids, site numbers, text offsets) as a host array, downloaded on first use."""
    if self._asm is None:
        self._asm = self.be.download(self._d_asm, np.int64, ASM_FIELDS * self.n_nodes).reshape(self.n_nodes, ASM_FIELDS)
        # job offsets live in the scan scratch; A_JOB only says who wrote the text (-1: host)
    return self._asm


ForestEngine.assemble_prgs = assemble_prgs
ForestEngine.asm = property(_asm_host)
ForestEngine.node_id = property(lambda self: self.asm[:, A_PRE])


def _site_count(self: ForestEngine) -> np.ndarray:
    if self._site_count is None:
        self._site_count = self.be.download(self._d_nsites, np.int64, len(self._msas))
    return self._site_count


ForestEngine.site_count = property(_site_count)
ForestEngine.tree_sizes = property(lambda self: np.where(self.root_of >= 0, self.asm[np.maximum(self.root_of, 0), A_SIZE], 0))


def forest_stored_alignments(self: ForestEngine, node_idx: np.ndarray) -> List[np.ndarray]:
    """node.alignment of the listed nodes as cell codes: the node's rows x columns without the columns that hold only gaps
    (recursion_tree.py:45 -> utils/seq_utils.py:193-216), compacted ON THE DEVICE: mprg_column_masks over the nodes' views,
    then mprg_compact_columns (A8); one download of the dense matrices."""
    be, t = self.be, self.tab
    node_idx = np.asarray(node_idx, np.int64)
    n = len(node_idx)
    if n == 0:
        return []
    meta = self.meta_arr[t["msa"][node_idx]]
    S, C = t["nrows"][node_idx], t["ncols"][node_idx]
    tab = np.zeros((n, VF), np.int64)
    tab[:, 0:4] = meta[:, 0:4]
    tab[:, 4], tab[:, 5], tab[:, 6], tab[:, 7] = t["rows_off"][node_idx], S, t["col0"][node_idx], C
    tab[:, 8], tab[:, 9] = np.cumsum(C) - C, np.cumsum(S) - S
    total_cols = int(C.sum())
    out_off = np.cumsum(S * C) - S * C
    out_bytes = int((S * C).sum())
    d_views = be.upload(tab)
    work, rpc = self._mask_work(tab)
    d_work, d_mask = be.upload(work), be.zeros(4 * max(total_cols, 1))
    cells = float((S * C).sum())
    be.call("mprg_column_masks", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(self.d_pool), be.ptr(d_work), work.shape[0], rpc,
            be.ptr(d_mask), be.stream, work=cells)
    chunk = 64
    wr = self._row_chunk_work(tab, chunk)
    d_wr, d_out, d_off, d_kept = be.upload(wr), be.empty(max(out_bytes, 16)), be.upload(out_off), be.zeros(4 * n)
    be.call("mprg_compact_columns", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(self.d_pool), be.ptr(d_wr), len(wr), chunk,
            be.ptr(d_mask), be.ptr(d_out), be.ptr(d_off), be.ptr(d_kept), be.stream, work=2 * cells)
    kept = be.download(d_kept, np.int32, n)
    dense = be.download(d_out, np.uint8, out_bytes)
    return [dense[o:o + s_ * k].reshape(s_, k) for o, s_, k in zip(out_off.tolist(), S.tolist(), kept.tolist())]


def forest_tree_dump(self: ForestEngine, mi: int, ids: List[str]) -> list:
    """Preorder dump of one tree (same shape as oracle.from_msa_oracle.tree_dump); requires assemble_prgs() first."""
    t = self.tab
    node_id = self.node_id
    kinds = {KIND_LEAF: "leaf", KIND_INTERVAL: "interval", KIND_CLUSTER: "cluster"}
    if getattr(self, "_pool_cache_used", -1) != self.pool_used:
        self._pool_cache, self._pool_cache_used = self.pool_host(), self.pool_used
    pool = self._pool_cache
    order = []
    stack = [int(self.root_of[mi])]
    while stack:
        ni = stack.pop()
        order.append(ni)
        stack.extend(reversed([int(t["first_child"][ni]) + j for j in range(int(t["n_child"][ni]))]))
    blocks = self.stored_alignments(np.asarray(order))          # all-gap columns are not stored (recursion_tree.py:45)
    n_rows_msa = int(self.meta_arr[mi, 4])
    out = []
    for ni, block in zip(order, blocks):
        rows = self.node_rows(ni, pool)
        if rows is None:
            rows = np.arange(n_rows_msa)
        kids = [int(t["first_child"][ni]) + j for j in range(int(t["n_child"][ni]))]
        par = int(t["parent"][ni])
        out.append(dict(id=int(node_id[ni]), kind=kinds[int(t["kind"][ni])], level=int(t["level"][ni]),
                        parent=None if par < 0 else int(node_id[par]),
                        rows=[[ids[r], b.tobytes().decode()] for r, b in zip(rows, decode(block))],
                        children=[int(node_id[c]) for c in kids]))
    return out


def forest_prg_index_entries(self: ForestEngine, mi: int) -> np.ndarray:
    """[[start, end, node id], ...] of one alignment, device entries in preorder then those of host-expanded leaves
    (assemble_prgs(want_index=True) first, and collect its text: the entries arrive with it)."""
    ix, b = self._index
    dev = ix[b[mi]:b[mi + 1]]
    extra = self._host_index.get(mi)
    return dev if not extra else np.concatenate([dev, np.asarray(extra, np.int32).reshape(-1, 3)])


def forest_prg_index(self: ForestEngine, mi: int) -> list:
    """[[start, end, node_id], ...] sorted, for one alignment."""
    return sorted(forest_prg_index_entries(self, mi).tolist())


ForestEngine.prg_index_entries = forest_prg_index_entries
ForestEngine.stored_alignments = forest_stored_alignments
ForestEngine.tree_dump = forest_tree_dump
ForestEngine.prg_index = forest_prg_index


# ======================================================================================================= batch ingest
class _ArenaAlignment:
    """One alignment of a raw arena (the native batch parser's output): what the engine reads of an MSA — its ASCII matrix."""
    __slots__ = ("data", "pending_n")

    def __init__(self, data):
        self.data, self.pending_n = data, False


class _ArenaBatch:
    """The alignments of a raw arena as a sequence (len / index); matrices are views, made on demand."""

    def __init__(self, arena: np.ndarray, raw_off: np.ndarray, rows: np.ndarray, cols: np.ndarray):
        self.arena, self.raw_off, self.rows, self.cols = arena, raw_off, rows, cols

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        i = int(i)
        o, S, C = int(self.raw_off[i]), int(self.rows[i]), int(self.cols[i])
        return _ArenaAlignment(self.arena[o:o + S * C].reshape(S, C))


def load_raw(self: ForestEngine, arena_buf, arena: np.ndarray, raw_off: np.ndarray, rows: np.ndarray, cols: np.ndarray,
             has_n: Optional[np.ndarray] = None, ids_of=None):
    """Ingest of a batch the native parser laid out (mprg_ingest_fill_host): alignment i = rows[i] x cols[i] upper-cased ASCII
    bytes at arena[raw_off[i]:]; arena_buf is the backend's (pinned) buffer behind `arena`.  The layout tables are built
    array-at-a-time, the bytes go up in ONE copy from pinned memory, mprg_ingest codes and transposes on the device.
    has_n: alignments that still hold N (their load-time consensus needs the device's column counts first).
    ids_of(i): the row ids of alignment i (only asked for error messages and tree dumps)."""
    be = self.be
    M = len(rows)
    rows, cols, raw_off = rows.astype(np.int64), cols.astype(np.int64), raw_off.astype(np.int64)
    batch = _ArenaBatch(arena, raw_off, rows, cols)
    self._msas, self._ids_of = batch, ids_of
    self._ids = None
    if has_n is not None and has_n.any():          # utils/seq_utils.py:246-290 on the device counts; patches the arena in place
        todo = []
        for i in np.nonzero(has_n)[0].tolist():
            a = batch[i]
            a.pending_n = True
            todo.append(a)
        self._resolve_pending_n(todo)
    from .engine import _LazyCodes
    self.codes = _LazyCodes(batch)
    self.bad = {}
    al = lambda x, a: (x + a - 1) // a * a
    C0 = np.where(rows == 0, 0, cols)
    pitchC, pitchS = al(np.maximum(C0, 1), 16), al(np.maximum(rows, 1), 16)
    sz_rm, sz_cm = al(rows * pitchC, 256), al(C0 * pitchS, 256)
    ends = np.cumsum(sz_rm + sz_cm)
    rm = ends - sz_rm - sz_cm
    cm = rm + sz_rm
    from .engine import INGEST_TILE as T_
    tiles = ((rows + T_ - 1) // T_) * ((C0 + T_ - 1) // T_)
    itab = np.zeros((max(M, 1), 9), np.int64)
    itab[:M, 0], itab[:M, 1], itab[:M, 2], itab[:M, 3], itab[:M, 4] = raw_off, rows, C0, rm, cm
    itab[:M, 5], itab[:M, 6], itab[:M, 7], itab[:M, 8] = pitchC, pitchS, -1, np.cumsum(tiles) - tiles
    self.meta = np.stack([rm, cm, pitchC, pitchS, rows, C0], axis=1)
    arena_bytes = int(ends[-1]) + 256 if M else 256
    raw_bytes = int((raw_off + rows * cols).max()) if M else 0
    self.d_arena = be.empty(arena_bytes)
    d_raw = be.upload_from(arena_buf, raw_bytes)
    d_itab, d_status = be.upload(itab), be.empty(4 * max(M, 1))
    be.call("mprg_ingest", be.ptr(d_raw), be.ptr(d_itab), M, int(tiles.sum()), None, be.ptr(self.d_arena), arena_bytes,
            be.ptr(d_status), be.stream, work=3.0 * float((rows * cols).sum()))
    status = be.download(d_status, np.int32, M) if M else np.zeros(0, np.int32)
    from .msa import encode
    for i in np.nonzero(status)[0].tolist():          # utils/seq_utils.py:96-104: such a locus ends in SequenceCurationError
        data = batch[i].data
        r, c = np.argwhere(encode(data) == 255)[0]
        rid = ids_of(i)[r] if ids_of is not None else f"row {r}"
        self.bad[i] = SequenceCurationError(f"A slice of a sequence has a disallowed base ({chr(data[r, c])!r} in {rid}). Redo sequence curation.")
    self.counters["arena_bytes"] = arena_bytes


ForestEngine.load_raw = load_raw
