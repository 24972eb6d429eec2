/*
 * mprg.h — C ABI of libmprg_hip.so: the MI355X (gfx950) kernels of the `from_msa` PRG-construction hot path.
 *
 * The reference (iqbal-lab-org/make_prg v0.5.0) is pure Python and has no FFI; each entry point below replaces the
 * body of one (or a fused group of) reference function(s), cited as file:line relative to /root/reference/make_prg/.
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  All pointers are DEVICE pointers owned by the caller (the Python host
 *     allocates them as PyTorch-ROCm tensors) unless the name ends in _host.
 *   - Every function takes the hipStream_t to enqueue on (as void*), enqueues asynchronously and never
 *     synchronises.  Return value: 0 = ok, negative = error (mprg_last_error() gives text).
 *   - A call processes a BATCH of node views (one recursion level of many MSAs): one launch per level.
 *
 * Data layout
 *   - Cells are 1-byte codes: 0 A, 1 C, 2 G, 3 T, 4 '-', 5 R, 6 Y, 7 K, 8 M, 9 S, 10 W, 11 N.
 *   - `arena` holds every MSA twice: row-major (row r, column c at rm_base + r*pitchC + c; pitchC % 16 == 0) and
 *     transposed (cm_base + c*pitchS + r; pitchS % 16 == 0).  Column-parallel kernels read the first, row-parallel
 *     kernels the second, so both kinds of access are coalesced.
 *   - A view is MPRG_VIEW_FIELDS int64 values (see enum): a row subset (identity or an index list in `rowidx`) and a
 *     closed-open column range of one MSA.  col_off / row_off are exclusive prefix sums over the batch and index
 *     the per-column / per-row outputs.
 */
#ifndef MPRG_H
#define MPRG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MPRG_V_RM_BASE = 0, MPRG_V_CM_BASE = 1, MPRG_V_PITCH_C = 2, MPRG_V_PITCH_S = 3, MPRG_V_ROWS_OFF = 4,
  MPRG_V_N_ROWS = 5, MPRG_V_COL0 = 6, MPRG_V_N_COLS = 7, MPRG_V_COL_OFF = 8, MPRG_V_ROW_OFF = 9,
  MPRG_V_AUX0 = 10, MPRG_V_AUX1 = 11, MPRG_VIEW_FIELDS = 12
};
enum { MPRG_CODE_GAP = 4, MPRG_CODE_N = 11, MPRG_N_CODES = 12 };
enum { MPRG_IV_MATCH = 0, MPRG_IV_NONMATCH = 1 /* bit 0 of a triple's type word */,
       MPRG_IV_PURE = 2 /* bit 1: match interval straight from the scan, view without N / ambiguity codes */ };
/* per-view status bits written by mprg_partition */
enum { MPRG_ST_PARTITION_ERROR = 1, MPRG_ST_ALL_N_SLICE = 2 };
/* per-fit status bits written by the KMeans kernels */
enum { MPRG_KM_RELOCATED = 1 /* an empty cluster was relocated (info) */, MPRG_KM_UNSUPPORTED = 2 /* error */ };

const char *mprg_version(void);
const char *mprg_last_error(void);
/* number of compute units of the current device (grid sizing); <0 on error */
int mprg_device_cus(void);

/* A0 on the device — utils/io_utils.py:17-49 (load_alignment_file) / SURVEY.md §8(f)-2, (f)-4.
 * mprg_ingest: the batch's alignments as the parser produced them (`raw`: per alignment rows x columns ASCII bytes, no
 * padding) -> the arena's row-major and transposed coded copies (layout above; padding cells = 15).  Lower case folds to
 * upper case; a byte outside ACGT-RYKMSWN sets status[alignment] = 1 (the reference ends such a locus with
 * SequenceCurationError); N is replaced by n_replacement[repl_off + column] (a cell code) where the table gives an offset.
 * msa_table: n_msas x MPRG_I_FIELDS int64 {raw offset, rows, columns, row-major base, transposed base, pitchC, pitchS,
 * replacement offset (-1: keep N), first tile}; an alignment owns ceil(rows/64) * ceil(columns/64) consecutive tiles,
 * n_tiles in all.  arena_bytes of `arena` are initialised by the call.
 * mprg_column_residue_counts: for the load-time majority consensus (utils/seq_utils.py:246-290): per column of the listed
 * alignments the number of rows holding each of A C G T R Y K M S W and the first such row.  table: 4 int64 per alignment
 * {raw offset, rows, columns, col_off}; work: n_work x 2 int32 {table row, 256-column tile}; out: 20 int32 per column at
 * 20 * (col_off + c) = count[10], first_row[10] (0x7fffffff: absent).  The seeded random choice stays on the host. */
enum { MPRG_I_FIELDS = 9 };
int mprg_ingest(const uint8_t *raw, const int64_t *msa_table, int n_msas, int64_t n_tiles, const uint8_t *n_replacement,
                uint8_t *arena, int64_t arena_bytes, int32_t *status, void *stream);
int mprg_column_residue_counts(const uint8_t *raw, const int64_t *table, const int32_t *work, int n_work, int32_t *out,
                               void *stream);

/* A2 + A8 — utils/seq_utils.py:219-239 (get_consensus_from_MSA) and :193-216 (all-gap columns).
 * work: n_items x 3 int32 {view, first column of a 1024-column tile (relative to the view, multiple of 4 in
 * absolute arena columns), first row position of a row chunk}; rows_per_chunk rows per item.
 * out_mask[col_off + c] (uint32, must be zeroed) receives the OR over the view's rows of (1 << code). */
int mprg_column_masks(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int32_t *work,
                      int n_items, int rows_per_chunk, uint32_t *out_mask, void *stream);

/* A3-A6 — from_msa/interval_partition.py:81-252 (IntervalPartitioner) with utils/seq_utils.py:37-42
 * (has_empty_sequence) and the <2-sequences test of :187-217.  Gap runs: one workgroup per (view, 256-row chunk)
 * — work_rows: n x 2 int32 {view, chunk} —; the interval scan itself: one workgroup per view.
 * in:  mask (from mprg_column_masks), min_match_length.
 * scratch: maxrun uint32[total_cols] (zeroed), stack int32[4*total_cols], ivflag int32[total_cols*2] (zeroed)
 * out: iv int32[3*total_cols] as {start, stop, type} triples at 3*col_off, n_iv int32[n_views],
 *      status int32[n_views].
 * optional (all three NULL or all three set): view_out int32[8*n_views] = {n_iv, status, type of the first interval,
 *      flags (1: some column is not one plain base, i.e. the consensus has a '*' or a gap; 2: N or ambiguity codes
 *      occur), first triple of this view in iv_packed, 0, 0, 0}; iv_packed int32[3*total_cols] = the triples of all
 *      views back to back (a view's triples are contiguous; views in index order: their places are the exclusive prefix
 *      sum of n_iv, laid out by two small launches after the partition kernels); iv_count int32[1] = triples in the list.
 * fused_list / other_list (both NULL, or int32 device lists that together hold 0..n_views-1): the views of fused_list
 *      are SMALL — at most 512 rows and 1024 columns, rows x pitch <= 8192 bytes (pitch = columns rounded up to 4, +4
 *      if that is an even number of words), columns / max(min_match_length - 1, 1) + 4 <= 128 — and are handled by one
 *      workgroup each that holds the view's cells in LDS and computes their column masks and gap runs itself: they
 *      need NO work items in mprg_column_masks, none in work_rows here, and no scratch. */
int mprg_partition(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views,
                   const uint32_t *mask, int min_match_length, const int32_t *work_rows, int n_work_rows,
                   uint32_t *maxrun, int32_t *stack, int32_t *ivflag, int32_t *iv, int32_t *n_iv, int32_t *status,
                   int32_t *view_out, int32_t *iv_packed, int32_t *iv_count, const int32_t *fused_list, int n_fused,
                   const int32_t *other_list, int n_other, void *stream);

/* A9a/A13/A16 — from_msa/cluster_sequences.py:220-233 (ungap, group identical rows in first-appearance order),
 * utils/seq_utils.py:58-70 (unique gapped / ungapped counts).  Ungap + hash: one workgroup per (view, 256-row chunk)
 * — work_rows: n x 2 int32 {view, chunk} —, one wavefront per row; grouping: one workgroup per view.
 * views[AUX0] = byte offset (a multiple of 16) of this view's region in `ucodes`: n_rows * upitch bytes, upitch =
 * round_up(n_cols, 16); ungapped codes are stored ROW-MAJOR: character j of row position i at i*upitch + j.
 * per row (at row_off): ulen, rep_u (smallest row position with identical ungapped content), rep_g (same for gapped
 * content), d_of_row (index of the row's sequence among the distinct sequences of length >= kmer_size, first-appearance
 * order; -1 if shorter), s_of_row (index among the distinct shorter sequences; -1 if long).
 * per view, compact lists (at row_off): reps_pos / reps_len = row positions and ungapped lengths of the distinct rows;
 * seqrow = row positions of the distinct long sequences; occ_off (at row_off + view index, D+1 entries) = exclusive
 * prefix sums of their k-mer occurrence counts.
 * summary int64[8*n_views] = {distinct ungapped, distinct gapped, D (distinct long), T (k-mer occurrences),
 * total ungapped length of the distinct rows, distinct short, 0, 0}.  scratch: hashes uint64[2*total_rows].  * gcodes (optional, same size and layout as ucodes): receives a dense copy of every view's GAPPED rows (row i of a view at
 * aux0 + i * pitch, pitch = columns rounded up to 16); the gapped comparison of this call and mprg_cluster_further read it
 * instead of the arena when given. */
int mprg_ungap_dedupe(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views, int kmer_size,
                      const int32_t *work_rows, int n_work_rows, uint8_t *ucodes, uint64_t *hashes, int32_t *ulen,
                      int32_t *rep_u, int32_t *rep_g, int32_t *d_of_row, int32_t *s_of_row, int32_t *reps_pos,
                      int32_t *reps_len, int32_t *seqrow, int64_t *occ_off, int64_t *summary, uint8_t *gcodes, void *stream);

/* A9b — from_msa/cluster_sequences.py:26-38 (count_distinct_kmers): k-mer dictionary in first-appearance order.
 * One workgroup per clustering problem.  prob: n_probs x MPRG_PROB_FIELDS int64 (see enum).  seqrow int32[]:
 * row positions (within the view) of the D distinct long sequences in first-appearance order.
 * table: uint64 keys[cap] then uint32 minocc[cap], uint32 id[cap] per problem (cap = power of two >= 2*T).
 * out_V[n_probs] = number of distinct k-mers. */
enum {
  MPRG_P_VIEW = 0, MPRG_P_D = 1, MPRG_P_SEQROW_OFF = 2, MPRG_P_T = 3, MPRG_P_TABLE_OFF = 4, MPRG_P_TABLE_CAP = 5,
  MPRG_P_OCC_OFF = 6, MPRG_P_V = 7, MPRG_P_X_OFF = 8, MPRG_P_WS_OFF = 9, MPRG_P_LABEL_OFF = 10, MPRG_P_FLAG_OFF = 11,
  MPRG_PROB_FIELDS = 12
};
int mprg_kmer_dictionary(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size,
                         const uint8_t *ucodes, const int32_t *ulen, const int32_t *seqrow, int64_t *occ_off,
                         uint8_t *table, uint8_t *first_flag, int32_t *out_V, void *stream);
/* A9c — cluster_sequences.py:41-56 (count_kmer_occurrences): dense D x V count matrix (float64, zeroed by caller)
 * at prob[X_OFF] (in doubles) inside `xcounts`. */
int mprg_kmer_counts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                     const int32_t *ulen, const int32_t *seqrow, const int64_t *occ_off, const uint8_t *table,
                     double *xcounts, void *stream);

/* A11 — the reference's KMeans(n_clusters=k, random_state=2, algorithm="elkan").fit(X).predict(X)
 * (cluster_sequences.py:262-266; arithmetic restated from scikit-learn, see oracle/kmeans_oracle.c) with
 * n_init restarts.  Three launches:
 *   mprg_kmeans_prepare : per problem, centre X (prob[WS_OFF] workspace), row norms, tolerance.   (once per problem)
 *   mprg_kmeans_restarts: per fit = (problem, k): its n_init restarts side by side: k-means++ + Elkan iterations.
 *   mprg_kmeans_select  : per fit: best restart by the reference's rule, then predict().
 * A fit is 5 int32 (kinfo): {row of `prob`, k, first restart slot of the fit in the problem's workspace, offset (in
 * doubles) of this k's uniforms in `uniforms_dev`, offset added to prob[LABEL_OFF] for the fit's labels}; several k of
 * one problem can be fitted in one launch (they use disjoint restart slots), which the host uses to run the
 * reference's k = 2,3,4,... loop a few k at a time.
 * uniforms of one k: n_init * (1 + (k-1)*(2+int(ln k))) doubles of numpy RandomState(2).random_sample.
 * labels int32; km_status int32[n_fits] (MPRG_KM_*); km_info double[8*n_fits] = {inertia, n_iter of the best restart,
 * best restart, n distinct labels, total Elkan iterations, -, -, -}.
 * Workspace size per problem (doubles): mprg_kmeans_workspace_doubles(D, V, k_max, restart slots); -1 if k_max > 10 or
 * V > 4 194 304 features (the limits of the kernels). */
int64_t mprg_kmeans_workspace_doubles(int64_t D, int64_t V, int k_max, int n_restart_slots);
/* mprg_kmeans_prepare, problem lists (both NULL: every problem takes the global-memory form): the problems of lds_list
 * (int32 rows of `prob`) stage their matrix in LDS — lds_bytes >= 8 * (D * (V | 1) + 2 * V) for each of them, at most
 * MPRG_KMEANS_PREPARE_LDS_MAX — so that the ordered reductions read LDS; other_list takes the rest. */
enum { MPRG_KMEANS_PREPARE_LDS_MAX = 156 * 1024 };      /* gfx950: 160 KB of LDS per CU */
int mprg_kmeans_prepare(const int64_t *prob, int n_probs, const double *xcounts, double *ws, const int32_t *lds_list,
                        int n_lds, int64_t lds_bytes, const int32_t *other_list, int n_other, void *stream);
/* mprg_kmeans_restarts, xcounts: the count matrices mprg_kmer_counts wrote (prob[X_OFF]).  A fit whose counts fit a byte
 * and (with the feature means) 12 KB of LDS keeps them there and forms the centred values on the fly — the same
 * subtraction mprg_kmeans_prepare performed; NULL: every fit reads the centred matrix of the workspace.  Same results
 * either way.  n_init <= 16.  mprg_kmeans_prepare may be called several times with disjoint lists (n_probs = n_lds +
 * n_other of that call): hosts launch the problems in classes of LDS need, because every workgroup of a launch
 * allocates lds_bytes. */
int mprg_kmeans_restarts(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *uniforms_dev,
                         const double *xcounts, double *ws, int32_t *km_status, void *stream);
int mprg_kmeans_select(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *xcounts,
                       double *ws, int32_t *labels, double *km_info, void *stream);
/* A11, restarts + select as ONE launch.  slot_ws == NULL (the form the batch host uses): one workgroup per fit, restart
 * regions in the problems' workspaces exactly as for mprg_kmeans_restarts (kinfo field 2 honoured), next_fit unused.
 * slot_ws != NULL: persistent workgroups: at most n_slots workgroups, each
 * taking the next unclaimed fit (next_fit: one int32 of device scratch, zeroed by the call; list the biggest fits first)
 * and keeping the per-restart arrays of its current fit in its own scratch slot
 * (slot_ws + b * slot_stride_doubles; slot_stride_doubles >= n_init * mprg_kmeans_workspace_doubles' per-restart part for
 * every fit of the launch, i.e. (mprg_kmeans_workspace_doubles(D, V, k_max, n_init) - mprg_kmeans_workspace_doubles(D, V,
 * k_max, 0))).  The problems' workspaces (`ws`, prob[WS_OFF]) then only need their common part
 * (mprg_kmeans_workspace_doubles(D, V, k_max, 0)), written by mprg_kmeans_prepare and read-only here.  Outputs as
 * mprg_kmeans_restarts + mprg_kmeans_select; kinfo field 2 is ignored. */
int mprg_kmeans_fit(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *uniforms_dev,
                    const double *xcounts, double *ws, double *slot_ws, int64_t slot_stride_doubles, int n_slots,
                    int32_t *next_fit, int32_t *labels, double *km_info, int32_t *km_status, void *stream);
/* fills out[n] with numpy.random.RandomState(seed).random_sample(n) (host memory; MT19937) */
void mprg_random_sample_host(uint32_t seed, int n, double *out_host);

/* A10 — cluster_sequences.py:59-111 (majority string, Hamming distance, one-reference-like test, cluster_further).
 * Two launches: majority strings per (problem, 256-column tile) — work_cols: n x 2 int32 {problem, tile} — then
 * Hamming distances per (problem, 256-row chunk) — work_rows: n x 2 int32 {problem, chunk}.
 * A row takes part if d_of_row >= 0; its cluster is labels[prob[LABEL_OFF] + d_of_row] (labels == NULL: a single
 * cluster).  Ties in the per-column majority go to the symbol seen first in the order in which the reference enumerates
 * the cluster's rows (distinct sequence, then row).  If `assign` is given the labels of these problems are also copied
 * there (the fit is the accepted one).  out_further[n_probs] = 1 if some cluster is not one-reference-like.
 * km_info (optional, the km_info of the KMeans round these labels come from, problem p = fit p): lets the call follow
 * mprg_kmeans_fit without a host decision in between — a fit with fewer than k distinct labels is not accepted
 * (cluster_sequences.py:267-273: its labels are not copied to `assign`; the host ignores its out_further).
 * gcodes (optional): the dense gapped copies mprg_ungap_dedupe wrote for these views (same `views` table): the kernels then
 * read a view as one contiguous block instead of a narrow slice of every alignment row. */
int mprg_cluster_further(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                         int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                         const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                         int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, void *stream);

/* A12/A14 — cluster_sequences.py:287-296 + recursion_tree.py:558-572: row lists of the children of MultiClusterNodes.
 * split_info: n_probs x 3 int64 {number of KMeans clusters, offset of the problem's n_rows entries in pool_out,
 * offset of its child sizes in child_sizes}.  Children order: the cluster holding the first row, then the KMeans
 * clusters by label, then one cluster per distinct short sequence in first-appearance order; rows keep MSA order. */
int mprg_split_children(const int64_t *views, const int32_t *rowidx, const int64_t *prob, int n_probs,
                        const int64_t *split_info, const int32_t *d_of_row, const int32_t *s_of_row,
                        const int32_t *assign, int32_t *pool_out, int32_t *child_sizes, void *stream);

/* A16 (per leaf) — recursion_tree.py:266-300: expands leaves into allele copy jobs and writes the leaf's own site
 * markers.  leaves: n x 10 int64 {MSA row-major base, pitchC, rows_off (-1 identity), col0, ncols, reps_off into
 * reps_pos/reps_len (-1: one allele = the first row, length ncols), alleles, destination of the leaf text, site number
 * (0: single allele, no markers), first job index}.  jobs: 4 int64 per allele {source, columns, destination, length}. */
int mprg_leaf_jobs(const int64_t *leaves, int64_t n_leaves, const int32_t *rowidx, const int32_t *reps_pos,
                   const int32_t *reps_len, int64_t *jobs, uint8_t *out, void *stream);

/* A16 — recursion_tree.py:266-300 (leaf alleles into the PRG string).  jobs: n_jobs x 4 int64 {arena byte offset of
 * the first cell in the row-major copy, number of columns, destination offset in `out`, allele length}; each job writes the ASCII of
 * its non-gap cells.  Offsets come from the host's prefix sums over the recursion tree (site markers are written by
 * the host). */
int mprg_emit_alleles(const uint8_t *arena, const int64_t *jobs, int64_t n_jobs, uint8_t *out, void *stream);

/* (f)-1 output encoders, HOST functions (host pointers), one pass over a PRG string as PrgBuilder emits it.
 * reference make_prg/utils/prg_encoder.py:44-91 and make_prg/utils/gfa.py:16-109.
 * mprg_prg_encode_host: out[n] receives the uint32 stream (A C G T -> 1 2 3 4, markers as integers, the closing
 *   occurrence of an odd site marker as the even one); returns the count.
 * mprg_gfa_text_host: out[out_cap] receives the GFA1 text; returns its length, or MPRG_OUT_TOO_SMALL (-4): retry with a
 *   bigger buffer (256 + 48 n always suffices; 4096 + 3 n does for pan-genome PRGs).
 * Both return MPRG_NOT_PLAIN_STRING (-3) for anything PrgBuilder would not emit (the caller then uses the
 * reference-shaped slow path, which owns the reference's errors and its `str(site) in prg` substring test). */
enum { MPRG_NOT_PLAIN_STRING = -3, MPRG_OUT_TOO_SMALL = -4 };
long long mprg_prg_encode_host(const char *prg, long long n, uint32_t *out);
long long mprg_gfa_text_host(const char *prg, long long n, char *out, long long out_cap);

/* (f)-2 ingest, HOST functions: FASTA alignment text -> rows x columns matrix of upper-cased bytes, two passes.
 * reference utils/io_utils.py:17-31 (AlignIO.read(handle, "fasta") + upper-casing).  mprg_fasta_scan_host counts the
 * records and checks that all sequences have one length (0 ok, MPRG_NOT_PLAIN_STRING: bytes other than printable ASCII /
 * tab / CR / LF — the caller's Python parser takes those —, MPRG_RAGGED_ALIGNMENT: lengths differ);
 * mprg_fasta_fill_host writes matrix[r * seq_len + c] and, per record, the [start, end) offsets of its title in `text`. */
enum { MPRG_RAGGED_ALIGNMENT = -5 };
long long mprg_fasta_scan_host(const char *text, long long n, long long *n_records, long long *seq_len);
long long mprg_fasta_fill_host(const char *text, long long n, uint8_t *matrix, long long seq_len, long long *title_spans);

#ifdef __cplusplus
}
#endif
#endif
