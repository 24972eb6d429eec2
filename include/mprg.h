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
enum { MPRG_ST_PARTITION_ERROR = 1, MPRG_ST_ALL_N_SLICE = 2,
       MPRG_ST_BAD_LAUNCH = 4 /* a view of fused_list does not satisfy the fused launch shape: a host error, not a property of the data */ };
/* per-fit status bits written by the KMeans kernels */
enum { MPRG_KM_RELOCATED = 1 /* an empty cluster was relocated (info) */,
       MPRG_KM_UNSUPPORTED = 2 /* error: the relocation's selection ran out of frames (more than 5^10 samples) */ };

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
 * replacement offset (-1: keep N), first tile}; an alignment owns ceil(rows/T) * ceil(columns/T) consecutive tiles, T = MPRG_INGEST_TILE,
 * n_tiles in all.  arena_bytes of `arena` are initialised by the call.
 * mprg_column_residue_counts: for the load-time majority consensus (utils/seq_utils.py:246-290): per column of the listed
 * alignments the number of rows holding each of A C G T R Y K M S W and the first such row.  table: 4 int64 per alignment
 * {raw offset, rows, columns, col_off}; work: n_work x 2 int32 {table row, 256-column tile}; out: 20 int32 per column at
 * 20 * (col_off + c) = count[10], first_row[10] (0x7fffffff: absent).  The seeded random choice stays on the host. */
enum { MPRG_I_FIELDS = 9, MPRG_INGEST_TILE = 128 };
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

/* A8 — utils/seq_utils.py:193-216 (remove_columns_full_of_gaps_from_MSA; recursion_tree.py:45 stores the result as
 * node.alignment).  The all-gap columns of each view (mask == 1 << MPRG_CODE_GAP, from mprg_column_masks) are dropped, the rest
 * written as a dense n_rows x kept[view] matrix of cell codes, row-major, at out + out_off[view] (the caller lays `out` out for
 * n_rows x n_cols per view: kept is only known afterwards).  work: n_items x 2 int32 {view, row chunk}; rows_per_chunk rows per
 * item; kept: int32 per view. */
int mprg_compact_columns(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int32_t *work, int n_items,
                         int rows_per_chunk, const uint32_t *mask, uint8_t *out, const int64_t *out_off, int32_t *kept, void *stream);

/* A3-A6 — from_msa/interval_partition.py:81-252 (IntervalPartitioner) with utils/seq_utils.py:37-42
 * (has_empty_sequence) and the <2-sequences test of :187-217.  Gap runs: one workgroup per (view, 256-row chunk, 512-column
 * segment) — work_rows: n x 2 int32 {view, segment * row chunks of the view + row chunk}, ceil(rows / 256) * ceil(columns / 512) items
 * per view —; the interval scan itself: one workgroup per view.
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
 * utils/seq_utils.py:58-70 (unique gapped / ungapped counts).  Ungap + hash: one workgroup per (view, row chunk) — work_rows:
 * n x 2 int32 {view, chunk}; a chunk is 256 rows, 8 for a view of more than 4 096 columns —, one wavefront per row; grouping: one
 * workgroup per view, after a scan over the same work items for the views of more than 512 rows.
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
 * Keys: the k-mer packed at 4 bits per character for kmer_size <= 16; beyond, a seeded 64-bit hash whose every use is verified
 * character by character (a collision rebuilds the dictionary with the next seed), so ids are exact for any k-mer size.
 * out_V[n_probs] = number of distinct k-mers (bits 0-23, saturating at 0xffffff) | the hash seed that held << 24 (0x7f: none of 64 did).  The seed must
 * be passed on to mprg_kmer_counts in bits 40+ of prob[MPRG_P_TABLE_CAP]. */
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
/* mprg_kmer_dictionary with `parts` workgroups per problem (1..1024; k-mer sizes up to 16: packed keys) — four launches: clear, insert,
 * first-appearance flags + their count per part, ids.  Same table, flags and ids.  part_counts: int32 [n_probs * parts] of scratch.
 * For levels whose problems hold millions of k-mer occurrences (the top of one deep alignment: 2 x 10^8 through one workgroup otherwise). */
int mprg_kmer_dictionary_parts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                               const int32_t *ulen, const int32_t *seqrow, int64_t *occ_off, uint8_t *table, uint8_t *first_flag,
                               int32_t *out_V, int parts, int32_t *part_counts, void *stream);
/* the same with `parts` workgroups per problem sharing its occurrences (1..1024): for levels that hold a problem of millions of
 * k-mer occurrences (the top of one deep alignment), whose single workgroup would otherwise decide the launch's duration */
int mprg_kmer_counts_parts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                     const int32_t *ulen, const int32_t *seqrow, const int64_t *occ_off, const uint8_t *table,
                     double *xcounts, int parts, void *stream);

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
int mprg_kmeans_fit(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init,
                    const double *uniforms_dev, const double *xcounts, double *ws, double *slot_ws, int64_t slot_stride_doubles,
                    int n_slots, int32_t *next_fit, int32_t *labels, double *km_info, int32_t *km_status, void *stream);
/* A11, the WAVE form of mprg_kmeans_fit for small fits (the rule in pan-genome alignments: ~14 sequences x ~90 k-mers): one
 * wavefront per fit, its restarts one after the other with the running restart's whole state in a compact LDS region, the best
 * restart chosen incrementally, predict() at the end.  Same results as the other forms.  fit_list (optional, also in
 * mprg_kmeans_fit's one-workgroup-per-fit form): int32 rows of kinfo this launch handles (n_fits of them); outputs
 * (km_info, km_status) stay indexed by the kinfo row.  lds_class: mprg_kmeans_wave_class(D, V, k) of every fit of the launch
 * (0..3: 7.4 / 13.6 / 23.8 / 38.1 KB regions; -1: the fit needs the workgroup form).  The final centres of the best restart
 * pass through restart slot 0 of the problem's workspace. */
/* np.argpartition(values, kth) as scikit-learn's empty-cluster relocation calls it (_k_means_common.pyx:167-211, float64, no
 * NaN): the generic arg-introselect of the reference's locked NumPy 1.24 including its median-of-medians fallback.  perm
 * (int32 [n]) receives the permutation, ok[0] = 1 (0: more than 5^10 elements).  Exposed for its own parity test; the KMeans
 * kernels run the same code inside a fit. */
int mprg_argpartition(const double *values, int32_t *perm, int n, int kth, int32_t *ok, void *stream);
/* the SPLIT form of mprg_kmeans_fit for launches that would not fill the device: a 64-thread workgroup per RESTART (n_fits *
 * n_init of them, restart r on slot `first slot + r` of the problem's workspace), then the selection over the same fit list in a
 * second launch.  Same arithmetic and results as mprg_kmeans_fit; more total work (what a fit's restarts share is redone per
 * restart), a shorter launch when the device would otherwise idle. */
int mprg_kmeans_fit_split(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init,
                          const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                          int32_t *km_status, void *stream);
/* K6 for BIG problems, whose fits all take mprg_kmeans_fit_wide: statistics, centred matrix, norms — and
 *   xbytes (optional): the raw counts as BYTES, rows of pitch round_up(V, 4) made an odd number of 4-byte words, problem p at byte
 *     8 * prob[p][X_OFF] of a buffer as large as xcounts; what the wide fits stream instead of the centred doubles (an eighth of the bytes);
 *   with_tables = 0: the sample-sample tables of the seeding are NOT made (2.5 D^2 chains per problem) — the wide fits compute the few
 *     dozen rows they ask for; only mprg_kmeans_fit_wide may fit such a problem.  1: made, by 32 x 32 tiles of sample pairs staged through
 *     LDS; 2: made by a thread per table element (the form before round 5, kept for comparison: the same doubles).
 * list: rows of `prob` (NULL: 0..n-1). */
int mprg_kmeans_prepare_big(const int64_t *prob, const double *xcounts, double *ws, const int32_t *list, int n_list, uint8_t *xbytes,
                            int with_tables, void *stream);
/* mprg_kmeans_fit_split with a WIDE workgroup (1 024 threads) per restart: for BIG fits — hundreds to thousands of distinct sequences x
 * thousands of k-mers, the clustering problems of one deep alignment — whose phases are thousands of chains as long as the k-mer
 * dictionary: ten workgroups on ten CUs instead of ten restarts behind one CU's L1 (profiles/r04/deep_alignment.md).
 * xbytes (optional): the byte matrix mprg_kmeans_prepare_big wrote for these problems. */
int mprg_kmeans_fit_wide(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init,
                         const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                         int32_t *km_status, const uint8_t *xbytes, void *stream);
int mprg_kmeans_wave_class(int64_t D, int64_t V, int k);
/* A11, the workgroup form for SMALL fits: 128-thread workgroups with a trimmed static LDS (8 KB pool; small_class 0 also a
 * 6 x 6 centre-centre table, i.e. k <= 6), so that 6-8 fits are resident per CU instead of 4.  mprg_kmeans_small_class(D, V, k,
 * n_init): 0 / 1, or -1 if the fit needs the general form (mprg_kmeans_fit).  Arguments as mprg_kmeans_fit_wave. */
int mprg_kmeans_small_class(int64_t D, int64_t V, int k, int n_init);
int mprg_kmeans_fit_small(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int small_class, int n_init,
                          const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                          int32_t *km_status, void *stream);
/* A11, the LDS form for SMALL fits (round 6; the default for every fit it accepts): a workgroup per fit, its n_init restarts side by
 * side, with the state of ALL restarts (lower / upper bounds, the iteration's distances, labels, centre shifts and norms), the fit's
 * counts (bytes) and the k-means++ seeding's inputs (uniforms, row norms, the sample-sample tables where they fit) in LDS for the
 * whole fit — dynamic LDS sized by the fit's class —; only the centres stay in the problem's workspace.  Best restart and predict()
 * in the same workgroup.  mprg_kmeans_lds_class(D, V, k, n_init): 0..5 (13.5 / 20 / 33.5 / 46.75 / 73.5 / 128 KB of dynamic LDS: 8 / 6 / 4 / 3 / 2 / 1 workgroups per CU), or -1 if the fit
 * needs another form (more than 64 distinct sequences, more than 10 restarts, a state beyond the largest class, a count matrix beyond
 * workspace prepared WITHOUT the sample-sample tables: the fit reads them — mprg_kmeans_prepare_big with_tables = 0 leaves them out; the
 * forest's control steps look that up in the workspace themselves, a caller of this function must know).  Arguments and
 * results as mprg_kmeans_fit_small; every fit of a launch must be of class <= lds_class.
 * Replaces, for these fits, scikit-learn's KMeans.fit + predict behind cluster_sequences.py:262-266. */
int mprg_kmeans_lds_class(int64_t D, int64_t V, int k, int n_init);
int mprg_kmeans_fit_lds(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int lds_class, int n_init,
                        const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                        int32_t *km_status, void *stream);
int mprg_kmeans_fit_wave(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int lds_class, int n_init,
                         const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                         int32_t *km_status, void *stream);
/* fills out[n] with numpy.random.RandomState(seed).random_sample(n) (host memory; MT19937) */
void mprg_random_sample_host(uint32_t seed, int n, double *out_host);

/* A10 — cluster_sequences.py:59-111 (majority string, Hamming distance, one-reference-like test, cluster_further).
 * Two launches: majority strings per (problem, column tile) — work_cols: n x 2 int32 {problem, tile}; a tile is 256 columns, 32 for a
 * problem whose view has more than 1 024 rows (MPRG_CF_TILE / _TILE_BIG / _ROWS below) — then
 * Hamming distances per (problem, row chunk) — work_rows: n x 2 int32 {problem, chunk}; a chunk is 256 rows, 16 for a problem of
 * 512 columns or more and more than 1 024 rows (MPRG_CF_WIDE / _ROWS / _ROWS_WIDE).
 * A row takes part if d_of_row >= 0; its cluster is labels[prob[LABEL_OFF] + d_of_row] (labels == NULL: a single
 * cluster).  Ties in the per-column majority go to the symbol seen first in the order in which the reference enumerates
 * the cluster's rows (distinct sequence, then row).  If `assign` is given the labels of these problems are also copied
 * there (the fit is the accepted one).  out_further[n_probs] = 1 if some cluster is not one-reference-like.
 * km_info (optional, the km_info of the KMeans round these labels come from, problem p = fit p): lets the call follow
 * mprg_kmeans_fit without a host decision in between — a fit with fewer than k distinct labels is not accepted
 * (cluster_sequences.py:267-273: its labels are not copied to `assign`; the host ignores its out_further).
 * gcodes (optional): the dense gapped copies mprg_ungap_dedupe wrote for these views (same `views` table): the kernels then
 * read a view as one contiguous block instead of a narrow slice of every alignment row.
 * kinfo (optional, the fit descriptors of that round, problem p = fit p): a problem whose descriptor says k = 0 sat the round out
 * (mprg_forest_kloop_advance) and is skipped here too.  In mprg_kmeans_fit (slot_ws == NULL) a fit with k = 0 returns at once. */
enum { MPRG_CF_TILE = 256, MPRG_CF_TILE_BIG = 32, MPRG_CF_ROWS = 1024, MPRG_CF_WIDE = 512, MPRG_CF_ROWS_WIDE = 16 };
int mprg_cluster_further(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                         int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                         const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                         int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, const int32_t *kinfo,
                         void *stream);
/* mprg_cluster_further given the caller's bound on the rows of a problem's view (max_rows; 0: none known, as mprg_cluster_further):
 * problems of more than MPRG_CF_ROWS rows get their majority strings from a launch of wide workgroups over the same column tiles
 * (their slices of rows share the tile's counters in LDS), which is left out when the bound says no problem is that tall. */
int mprg_cluster_further_bounded(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                                 int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                                 const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                                 int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, const int32_t *kinfo,
                                 long long max_rows, void *stream);

/* A11 + A10 + A12 as ONE launch per workgroup form — cluster_sequences.py:256-274, the whole loop
 *     while cluster_further(...): num_clusters += 1; KMeans(num_clusters).fit(X).predict(X); accept / revert
 * of every problem: a problem's workgroup walks k = 2, 3, ... itself (fit = the bodies of mprg_kmeans_fit / _fit_small, best
 * restart + predict, cluster_further on the view's dense gapped copy, the reference's rules: fewer than k distinct labels ->
 * revert to k - 1 and stop; every cluster one-reference-like -> accept and stop; k > 10 or k == D -> stop), so a recursion
 * level costs one launch per form instead of ~4 launches per round k — the launch tails that decide small batches.
 * prob / xcounts / ws as for mprg_kmeans_fit (mprg_kmeans_prepare must have run); views = the problems' view table (prob[VIEW]),
 * d_of_row / gcodes from mprg_ungap_dedupe (gcodes is required), scratch as for mprg_cluster_further (3 int32 per column of the
 * views).  uniform_offsets_host: HOST array of 11 int32, [k] = offset (doubles) of k's uniforms in uniforms_dev, k = 2..10.
 * In/out per problem: num_clusters (start: 1), active (start: 1; 0 afterwards unless a later form must continue), labels /
 * assign (assign = labels of the last ACCEPTED fit), km_info / km_status of the problem's last fit.
 * stats (device int64[96], accumulated): 80 fits run, 85 / 93 algorithmic bytes 8 D V (iterations + n_init) of the fits run in the
 * general / small form (doubles), 84 cells visited by cluster_further (double), 82 set if a fit reported MPRG_KM_UNSUPPORTED.
 * forms: which workgroup forms this call launches, in this order on `stream`: MPRG_LOOP_GENERAL (256 threads; with
 * MPRG_LOOP_SKIP_SMALL it leaves the problems mprg_kmeans_small_class accepts to the small form), MPRG_LOOP_SMALL_LOW (128
 * threads, k = 2..6), MPRG_LOOP_SMALL_HIGH (128 threads, continues the small problems still active with k = 7..10; must follow
 * _LOW on the same stream).  The general and the small launches are independent: a host may put them on different streams. */
enum { MPRG_LOOP_GENERAL = 1, MPRG_LOOP_SMALL_LOW = 2, MPRG_LOOP_SMALL_HIGH = 4, MPRG_LOOP_SKIP_SMALL = 8,
       MPRG_LOOP_LDS = 16 /* round 6: the LDS form of the fit (mprg_kmeans_fit_lds) inside the loop, one launch per LDS class, BEFORE the other
                             forms of the call; rounds whose fit has no class are left to MPRG_LOOP_GENERAL (without MPRG_LOOP_SKIP_SMALL) */,
       MPRG_LOOP_SKIP_LDS = 32 /* with MPRG_LOOP_GENERAL: leave the problems whose NEXT round has an LDS class alone — the general form may then
                                  run beside the MPRG_LOOP_LDS launches (other stream); a last MPRG_LOOP_GENERAL call without it, after both,
                                  takes the problems that left the classes on the way */ };
int mprg_cluster_loop(const int64_t *views, const int64_t *prob, int n_probs, int n_init, const double *uniforms_dev,
                      const int32_t *uniform_offsets_host, const double *xcounts, double *ws, const int32_t *d_of_row,
                      const uint8_t *gcodes, int32_t *scratch, int32_t *labels, int32_t *assign, double *km_info, int32_t *km_status,
                      int32_t *num_clusters, int32_t *active, int64_t *stats, int forms, void *stream);

/* A12 — cluster_sequences.py:256-274, every round of the loop AT ONCE for levels that hold a few BIG problems: the KMeans fits of
 * different k share nothing but the problem's read-only data, so the general-form fit of every k = 2 .. min(10, D - 1) of every
 * problem can go out in ONE launch of mprg_kmeans_fit_wide (n_fits = 9 n_probs, fit_list = NULL) — ninety workgroups per problem
 * instead of ten, nine times over — and the rounds then settle in the reference's order on results that are already there.  This
 * writes the launch's table: kinfo_out int32 [9 n_probs][5], row (k - 2) n_probs + b = {b, k (0: round k's fit of this problem
 * takes another form, or k >= D), restart slot (k - 2) n_init, uniform_offsets_host[k], (k - 2) labels_per_k}.  The problems'
 * workspaces must hold 9 n_init restart slots (mprg_kmeans_workspace_doubles), labels / km_info / km_status nine slices of
 * labels_per_k / 8 n_probs / n_probs entries: slice k - 2 is what round k's other launches and mprg_cluster_further are given.
 * mode: MPRG_F_KM_MODE (which forms small fits take).  ws (round 6; may be NULL): the problems' workspaces AFTER mprg_kmeans_prepare* —
 * the LDS form (mode bit 2) takes no fit of a problem that was prepared without the seeding's tables. */
int mprg_kmeans_speculative_kinfo(const int64_t *prob, int n_probs, int n_init, int mode, const int32_t *uniform_offsets_host,
                                  long long labels_per_k, int32_t *kinfo_out, const double *ws, void *stream);

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

/* ---- the recursion forest on the device (NodeFactory.build's decisions and children,
 * recursion_tree.py:401-471; the k = 2..10 loop's control, cluster_sequences.py:256-274; node ids and PRG layout,
 * recursion_tree.py:48-55, :194-300, prg_builder.py:100-119).
 * The node table, the view / problem / work-item tables of every launch and the clustering loop's state live in device
 * memory; each step is "count per item -> exclusive prefix sums -> fill".  The host (which still drives the recursion and owns
 * every buffer) reads one small header of totals per step to size the next buffers.
 * node table: MPRG_NODE_FIELDS int64 per node; the nodes of a recursion level are one contiguous range, the children of a node
 * are contiguous inside the next level's range.
 * F (every entry point's first argument): a HOST array of MPRG_F_FIELDS int64 — device addresses and sizes, read at call time.
 * MPRG_F_VALS: int64 [MPRG_FOREST_VALS_COLS][items] scratch of the step (column-major); MPRG_F_SCAN_TMP: int64 [items / 2048 + 2]
 * [MPRG_FOREST_VALS_COLS];
 * MPRG_F_HDR: int64 [MPRG_FOREST_HDR] of device memory that receives the step's totals (column sums, see each step). */
enum {
  MPRG_N_MSA = 0, MPRG_N_PARENT = 1, MPRG_N_LEVEL = 2 /* nesting level */, MPRG_N_ROWS_OFF = 3 /* row pool offset, -1: all rows */,
  MPRG_N_NROWS = 4, MPRG_N_COL0 = 5, MPRG_N_NCOLS = 6, MPRG_N_FLAGS = 7 /* MPRG_NF_* */, MPRG_N_KIND = 8, MPRG_N_FIRST_CHILD = 9,
  MPRG_N_NCHILD = 10, MPRG_N_LVL = 11 /* recursion level (breadth-first) that classified the node */,
  MPRG_N_REPS_OFF = 12 /* offset of its distinct rows in that level's reps_pos / reps_len, -1: one allele = its columns */,
  MPRG_N_NSEQ = 13 /* distinct ungapped rows */, MPRG_N_ACHARS = 14 /* their total length */, MPRG_N_AUX = 15, MPRG_NODE_FIELDS = 16
};
enum { MPRG_KIND_LEAF = 0, MPRG_KIND_INTERVAL = 1, MPRG_KIND_CLUSTER = 2 };
enum { MPRG_NF_PURE = 1 /* match interval of a view without N / ambiguity codes: a one-allele leaf, no kernel visits it */,
       MPRG_NF_SPECIAL = 2 /* its columns hold N or ambiguity codes */, MPRG_NF_FORCED = 4 /* tree root: never a cluster node */,
       MPRG_NF_CAND = 8, MPRG_NF_DLEAF = 16, MPRG_NF_PQ = 32 };
enum { MPRG_ASM_SIZE = 0, MPRG_ASM_PRE = 1 /* node id */, MPRG_ASM_SITE = 2, MPRG_ASM_TOTAL = 3, MPRG_ASM_START = 4,
       MPRG_ASM_NSEQ = 5, MPRG_ASM_ACHARS = 6, MPRG_ASM_JOB = 7, MPRG_ASM_FIELDS = 8 };
enum { MPRG_FOREST_VALS_COLS = 16, MPRG_FOREST_HDR = 96 };
enum {
  /* batch */
  MPRG_F_NODES = 0, MPRG_F_N_NODES = 1, MPRG_F_META = 2 /* int64 [alignments][6]: rm, cm, pitchC, pitchS, rows, columns */,
  MPRG_F_N_MSAS = 3, MPRG_F_FAILED = 4 /* int32 [alignments] */, MPRG_F_ERR_FIRST = 5 /* uint64 [alignments], all ones */,
  MPRG_F_POOL = 6 /* int32 row pool */, MPRG_F_POOL_USED = 7, MPRG_F_ARENA = 8, MPRG_F_MAX_NESTING = 9, MPRG_F_MIN_MATCH = 10,
  MPRG_F_FUSED_ENABLED = 11, MPRG_F_N_INIT = 12, MPRG_F_VALS = 13, MPRG_F_SCAN_TMP = 14, MPRG_F_HDR = 15,
  /* level */
  MPRG_F_F0 = 16, MPRG_F_N = 17, MPRG_F_LVL = 18, MPRG_F_VIEWS = 19, MPRG_F_VIEW2NODE = 20, MPRG_F_FUSED_LIST = 21,
  MPRG_F_OTHER_LIST = 22, MPRG_F_MASK_WORK = 23, MPRG_F_RPC_IDX = 24 /* rows per mask item = 1024 >> this */, MPRG_F_GAP_WORK = 25,
  MPRG_F_VIEW_OUT = 26, MPRG_F_IV_PACKED = 27, MPRG_F_N_VIEWS = 28,
  MPRG_F_SUB = 29, MPRG_F_SELNODE = 30, MPRG_F_DD_WORK = 31, MPRG_F_SUMMARY = 32, MPRG_F_NSEL = 33,
  MPRG_F_T1 = 34, MPRG_F_WORK_COLS = 35, MPRG_F_WORK_ROWS = 36, MPRG_F_FURTHER = 37, MPRG_F_NPQ = 38,
  MPRG_F_PTAB0 = 39, MPRG_F_PTAB = 40, MPRG_F_DV = 41, MPRG_F_P = 42, MPRG_F_CLS_LISTS = 43 /* int32 [5][P] */,
  MPRG_F_NUM_CLUSTERS = 44, MPRG_F_ACTIVE = 45, MPRG_F_KINFO = 46, MPRG_F_KM_INFO = 47, MPRG_F_KM_STATUS = 48,
  MPRG_F_SPT = 49, MPRG_F_SP = 50, MPRG_F_SPLITNODE = 51, MPRG_F_CHILD_SIZES = 52, MPRG_F_NSPLITS = 53,
  /* assembly */
  MPRG_F_ASM = 54, MPRG_F_ROOT_OF = 55, MPRG_F_SPECIAL_LIST = 56, MPRG_F_SPECIAL_CAP = 57, MPRG_F_PATCH = 58, MPRG_F_N_PATCH = 59,
  MPRG_F_LEVELS = 60 /* HOST int64 [levels][4]: first node, nodes, reps_pos, reps_len (device addresses or 0) */, MPRG_F_N_LEVELS = 61,
  MPRG_F_VALS_MSA = 62 /* int64 [alignments]: every alignment's first place in the preorder layouts */, MPRG_F_VALS_NODE = 63,
  MPRG_F_VALS_POS = 64, MPRG_F_N_SITES = 65, MPRG_F_JOBS = 66, MPRG_F_OUT = 67, MPRG_F_MSA_BASE = 68, MPRG_F_UOFF = 69 /* .. 79: offset of k's uniforms, k = 2..10 */,
  MPRG_F_FIT_LISTS = 81 /* int32 [7][P]: the round's fits per launch list (hdr 86-92) */,
  MPRG_F_INDEX_OUT = 83 /* optional int32 [jobs][3]: the PRG index {start, end, node id}, per locus contiguous */,
  MPRG_F_EX_RECORDS = 84, MPRG_F_EX_ROWS = 85,
  MPRG_F_KM_MODE = 82 /* bit 0: small fits in the wave form, bit 1: small fits in the small workgroup form */,
  MPRG_F_HDR_HOST = 80 /* optional: host-visible (pinned) int64 [MPRG_FOREST_HDR]; every step that fills MPRG_F_HDR copies it there */,
  /* mprg_forest_level only (below): the device state, the level's index, the buffers that the per-step hosts hand to the data entry
   * points directly, and the CAPACITIES of the level's buffers */
  MPRG_F_DS = 96, MPRG_F_LEVEL_INDEX = 97,
  MPRG_F_MASK = 98, MPRG_F_MAXRUN = 99, MPRG_F_STACK = 100, MPRG_F_IVFLAG = 101, MPRG_F_IV = 102, MPRG_F_NIV = 103, MPRG_F_STATUS = 104,
  MPRG_F_IVC = 105, MPRG_F_UCODES = 106, MPRG_F_GCODES = 107, MPRG_F_HASHES = 108, MPRG_F_ULEN = 109, MPRG_F_REP_U = 110, MPRG_F_REP_G = 111,
  MPRG_F_D_OF_ROW = 112, MPRG_F_S_OF_ROW = 113, MPRG_F_REPS_POS = 114, MPRG_F_REPS_LEN = 115, MPRG_F_SEQROW = 116, MPRG_F_OCC_OFF = 117,
  MPRG_F_CF_SCRATCH = 118, MPRG_F_TABLE = 119, MPRG_F_FLAG = 120, MPRG_F_X = 121, MPRG_F_WS = 122, MPRG_F_LABELS = 123, MPRG_F_ASSIGN = 124,
  MPRG_F_UNIFORMS = 125, MPRG_F_UOFF_HOST = 126 /* HOST int32 [11] */, MPRG_F_LOOP_FORMS = 127 /* MPRG_LOOP_* */,
  MPRG_F_SIDE_STREAM = 160 /* optional: a second stream; the small forms of the clustering loop run there beside the general form */,
  MPRG_F_MAX_ROWS = 161 /* optional: an upper bound on the rows of a view of this forest (the largest root; 0: unknown) — launches that only
                         * views of thousands of rows need are left out below it */,
  MPRG_F_CAP = 128 /* + MPRG_CAP_* */,
  MPRG_F_FIELDS = 192
};
enum {
  MPRG_CAP_TCOLS = 0 /* columns of the level's views */, MPRG_CAP_NFUSED = 1, MPRG_CAP_NOTHER = 2, MPRG_CAP_ITEMS = 3 /* mask work items at MPRG_F_RPC_IDX */,
  MPRG_CAP_NGAP = 4, MPRG_CAP_NODES = 5 /* rows of the node table */, MPRG_CAP_SROWS = 6 /* rows of the selected views */,
  MPRG_CAP_UBYTES = 7, MPRG_CAP_SCOLS = 8, MPRG_CAP_NDD = 9, MPRG_CAP_WC = 10, MPRG_CAP_WR = 11 /* work items of the k = 1 check */,
  MPRG_CAP_TABLE = 12, MPRG_CAP_FLAG = 13, MPRG_CAP_LO = 14, MPRG_CAP_XD = 15, MPRG_CAP_WSD = 16, MPRG_CAP_CLS = 17 /* .. 21: problems per
  mprg_kmeans_prepare class */, MPRG_CAP_LDS = 22 /* .. 25: LDS bytes each LDS class is launched with */, MPRG_CAP_NCHILD = 26,
  MPRG_CAP_POOL = 27 /* entries of the row pool */,
  MPRG_CAP_BIG = 28 /* optional (0: no limit): a clustering problem whose count matrix needs more bytes than this makes the level
                     * overflow (step MPRG_STEP_SIZES_SHAPE) — the host has wider launch forms for such problems (mprg_kmeans_fit_wide) */
};
/* ---- a recursion level WITHOUT a host wait.  mprg_forest_level enqueues every step of one level — S1 .. S7 below and the data entry
 * points between them (mprg_column_masks, mprg_partition, mprg_ungap_dedupe, mprg_cluster_further (k = 1), mprg_kmer_dictionary,
 * mprg_kmer_counts, mprg_kmeans_prepare, mprg_cluster_loop, mprg_split_children) — on `stream` and returns; no total is read back.
 * The host sizes the level's buffers from CAPACITIES it chooses (what the same level of a previous forest needed, plus headroom)
 * and passes them where the per-step entry points take exact counts: MPRG_F_N, _N_VIEWS, _NSEL, _NPQ, _P, _NSPLITS and
 * MPRG_F_CAP + MPRG_CAP_*.  Every launch is sized by its capacity; the exact counts are words of the device state that the count
 * steps write, workgroups and items beyond them return at once.  A total that exceeds its capacity sets ds[MPRG_DS_OVERFLOW]
 * (sticky) BEFORE anything is written beyond a buffer, and every later kernel of the forest returns at once: the host, which
 * looks at the device state once after the last level, then repeats the forest with the per-step entry points (exact sizes).
 * Device state MPRG_F_DS: int64 [MPRG_DS_GLOBAL + levels * MPRG_DS_LEVEL_WORDS], zeroed by the host, then ds[MPRG_DS_N] =
 * ds[MPRG_DS_NNODES] = number of roots.  Level l (MPRG_F_LEVEL_INDEX, counted by the host) owns six step blocks of
 * MPRG_FOREST_HDR words — block s at MPRG_DS_GLOBAL + (6 l + s) * MPRG_FOREST_HDR — that receive what the steps' headers hold
 * in the per-step form (frontier block: words 13 / 14 = first node / nodes of the level's frontier; sizes block: the
 * mprg_cluster_loop statistics, words 80 ..).  Buffers that must arrive zeroed: MPRG_F_MASK, _MAXRUN, _IVFLAG, _IVC, _X, _ASSIGN,
 * _KM_STATUS.  The clustering loop is the fused one (mprg_cluster_loop, forms MPRG_F_LOOP_FORMS). */
enum { MPRG_DS_OVERFLOW = 0 /* 0, or 100 * (level + 1) + the step whose totals did not fit */, MPRG_DS_F0 = 1, MPRG_DS_N = 2,
       MPRG_DS_NNODES = 3, MPRG_DS_POOL_USED = 4, MPRG_DS_LEVEL = 5 /* levels enqueued so far */,
       MPRG_DS_NFAILED = 6 /* set when a view's partition failed (its locus is dropped: MPRG_F_FAILED, MPRG_F_ERR_FIRST) */, MPRG_DS_GLOBAL = 16,
       MPRG_DS_LEVEL_WORDS = 6 * 96 };
enum { MPRG_STEP_FRONTIER = 0, MPRG_STEP_CLASSIFY = 1, MPRG_STEP_CLUSTER = 2, MPRG_STEP_PROBLEMS = 3, MPRG_STEP_SIZES = 4, MPRG_STEP_SPLITS = 5,
       /* overflow codes only: */ MPRG_STEP_SIZES_SHAPE = 7 /* an LDS class's launch size, or MPRG_CAP_BIG */, MPRG_STEP_BEGIN = 9 /* the frontier itself */ };
int mprg_forest_level(const int64_t *F, void *stream);
/* A level whose totals did not fit need not cost the forest: k_ds_begin keeps, in words 15-17 of every level's frontier block, what
 * the forest's state was when the level began (frontier size, node-table rows, row-pool entries).  mprg_forest_state_rewind puts the
 * device state back to the start of `level` and clears MPRG_DS_OVERFLOW; the host then enqueues that level and the ones after it
 * again with larger buffers (mprg_forest_level, same MPRG_F_LEVEL_INDEX) — the levels before it stand.  Safe because a level's kernels
 * return at once from the overflow on and its only write that a second run would read differently (the nesting level of a new
 * MultiClusterNode, k_sp_children) comes after the level's last capacity check. */
int mprg_forest_state_rewind(int64_t *ds, long long n_words, long long level, void *stream);
/* zeroes the device state (n_words int64) and sets the forest's roots: ds[MPRG_DS_N] = ds[MPRG_DS_NNODES] = n_roots */
int mprg_forest_state_init(int64_t *ds, long long n_words, long long n_roots, void *stream);
/* S1  frontier -> views.  hdr: 0 views, 1 their columns, 2 their rows, 3 fused views, 4 other views, 5-9 mask work items for
 *     row chunks of 1024/512/256/128/64 rows, 10 gap-run row chunks, 11 cells, 12 cells of the other views.  The host picks MPRG_F_RPC_IDX from 5-9, allocates
 *     and calls _fill: views, view2node, fused / other lists, work items of mprg_column_masks and mprg_partition. */
int mprg_forest_frontier_count(const int64_t *F, void *stream);
int mprg_forest_frontier_fill(const int64_t *F, void *stream);
/* S2  after mprg_partition (view_out, iv_packed): status -> failed loci; leaf / multi-interval / clustering candidate.
 *     hdr: 0 children of multi-interval nodes, 1 selected views (candidates + leaves with several distinct rows), 2 their rows,
 *     3 bytes of their ungapped rows, 4 their columns, 5 row chunks of mprg_ungap_dedupe, 6 their cells.  _children: the child nodes
 *     (at MPRG_F_N_NODES on), the selected views' table `sub`, selnode, the dedupe work items. */
int mprg_forest_classify(const int64_t *F, void *stream);
int mprg_forest_children(const int64_t *F, void *stream);
/* S3  after mprg_ungap_dedupe (summary): leaf alleles of the selected views; candidates that go on (recursion_tree.py:538-556).
 *     hdr: 0 problems of the k = 1 check, 1 their column tiles, 2 their row chunks, 3 cells of all candidates, 4 cells of the
 *     problems. */
int mprg_forest_cluster_count(const int64_t *F, void *stream);
int mprg_forest_cluster_fill(const int64_t *F, void *stream);
/* S4  after mprg_cluster_further(k = 1): hdr: 0 clustering problems, 1 bytes of k-mer tables, 2 bytes of first-appearance flags,
 *     3 distinct long sequences.  _fill: MPRG_F_PTAB0 (fields 0-6, 10, 11). */
int mprg_forest_problems_count(const int64_t *F, void *stream);
int mprg_forest_problems_fill(const int64_t *F, void *stream);
/* S5  after mprg_kmer_dictionary (MPRG_F_DV): hdr: 0 doubles of count matrices, 1 doubles of KMeans workspaces, 2-6 problems
 *     per mprg_kmeans_prepare class (LDS need <= 12, 24, 64, 156 KB, global form), 7 / 8 work items of mprg_cluster_further
 *     (column tiles, row chunks of ALL problems: MPRG_F_WORK_COLS / _ROWS are filled by _fill), 15 error: too many features, 16-20 largest
 *     LDS need per class.  _fill: MPRG_F_PTAB in launch order (biggest fits first), class lists, loop state. */
int mprg_forest_sizes_count(const int64_t *F, void *stream);
int mprg_forest_sizes_fill(const int64_t *F, void *stream);
/* S6  the clustering loop's control step before round k (k = 2 .. 11; cluster_sequences.py:256-274): settles round k-1 from
 *     km_info / km_status / out_further, writes the kinfo of round k (k = 0: the problem is done, its workgroups return).
 *     hdr (accumulated from mprg_forest_sizes_count on): 80 fits run, 81 / 85 / 93 KMeans algorithmic bytes of the fits run in
 *     the wave / general / small workgroup form (doubles), 82 unsupported fit,
 *     84 cells visited by the rounds' mprg_cluster_further (double); reset per call: 83 problems still active, 86-89 fits of
 *     round k for mprg_kmeans_fit_wave per LDS class, 90 fits for mprg_kmeans_fit, 91 / 92 for mprg_kmeans_fit_small class 0 / 1
 *     (listed in MPRG_F_FIT_LISTS; which forms are used: MPRG_F_KM_MODE — bit 0 the wave form, bit 1 the small workgroup form, bit 2
 *     (round 6, the hosts' default) the LDS form mprg_kmeans_fit_lds: its classes 0-3 take slots 86-89, classes 4 / 5 slots 91 / 92,
 *     and the algorithmic bytes of its fits are added to word 81). */
int mprg_forest_kloop_advance(const int64_t *F, int k, void *stream);
/* S7  after the loop: hdr: 0 new MultiClusterNodes, 1 their rows, 2 their children.  _fill: tables of mprg_split_children;
 *     _split_children (after it): the nodes become cluster nodes, their children are appended at MPRG_F_N_NODES. */
int mprg_forest_splits_count(const int64_t *F, void *stream);
int mprg_forest_splits_fill(const int64_t *F, void *stream);
int mprg_forest_split_children(const int64_t *F, void *stream);
/* KA  PRG assembly.  _special: hdr 0 = leaves the host must expand (N / ambiguity codes), listed with their node rows.
 *     _layout: node ids, site numbers, text lengths; hdr: 0 characters of all PRGs, 1 allele copy jobs.
 *     _emit: text starts, jobs, every marker; follow with mprg_emit_alleles(jobs). */
int mprg_forest_assemble_special(const int64_t *F, void *stream);
int mprg_forest_assemble_layout(const int64_t *F, void *stream);
int mprg_forest_assemble_emit(const int64_t *F, void *stream);
/* KE  export for the update data structure (after _layout): per node, at its preorder place (MPRG_F_VALS_MSA[alignment] + node
 *     id), 8 int32 {parent's node id, kind, nesting level, rows (-1: all rows of the alignment, -2: its parent's rows), first of
 *     its rows in the locus's slice of MPRG_F_EX_ROWS, first column, columns, alleles}; MPRG_F_EX_ROWS: the MSA rows of the nodes
 *     that own a list (children of cluster nodes), per locus contiguous.  _count: hdr 0 = entries of MPRG_F_EX_ROWS (uses
 *     MPRG_F_VALS_POS). */
int mprg_forest_export_count(const int64_t *F, void *stream);
int mprg_forest_export_fill(const int64_t *F, void *stream);
/* (f)-2 — the alignments of the exported loci at FOUR BITS per cell, for the update data structure (the reference pickles every
 * locus's PrgBuilder with its alignment, subcommands/from_msa.py:114-127; the cell codes 0..11 fit a nibble: half the bytes of the
 * ASCII matrix, which is 96 % of a member).  meta: MPRG_F_META's table (int64 [alignments][6]); row_base: int64 [n_msas + 1], the
 * first row of every alignment in the launch (total_rows = row_base[n_msas]); row r of alignment m goes to out_off[m] + r *
 * ceil(columns / 2): cell 2 j in the low nibble of byte j, 15 beside a last odd cell. */
int mprg_export_alignments(const uint8_t *arena, const int64_t *meta, const int64_t *row_base, const int64_t *out_off, long long n_msas,
                           long long total_rows, uint8_t *out, void *stream);

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

/* (f)-2 BATCH stages of the driver, HOST functions with their own threads (reference: one worker process per locus,
 * utils/io_utils.py:17-49, utils/input_output_files.py:73-162).
 * mprg_ingest_*: a list of FASTA files (paths: n_files NUL-terminated strings back to back) is read and scanned by n_threads
 * threads; info = 5 int64 per file {status (0, MPRG_NOT_PLAIN_STRING: gzip or bytes the Python parser takes,
 * MPRG_RAGGED_ALIGNMENT, MPRG_INGEST_UNREADABLE, MPRG_INGEST_NO_RECORDS), rows, columns, bytes of its titles joined by '\n',
 * flags (MPRG_INGEST_DUP_IDS, MPRG_INGEST_HAS_N)}; _fill writes file i's upper-cased matrix at arena + raw_off[i] (the caller's
 * pinned upload buffer; raw_off[i] < 0: skip) and its titles at titles + title_off[i].
 * mprg_encode_*: the batch's PRG text (one buffer; locus i = [base[i], base[i] + len[i]), len < 0: none) -> binary PRG words and
 * GFA bytes per locus (_sizes; -1: the one-pass encoders do not cover this string), then the encodings at the caller's offsets
 * and the CRC-32 of the three future zip members {PRG text, binary PRG, GFA text} per locus (_fill). */
enum { MPRG_INGEST_UNREADABLE_FILE = -6, MPRG_INGEST_NO_RECORDS_FILE = -7 };
enum { MPRG_INGEST_FLAG_DUP_IDS = 1, MPRG_INGEST_FLAG_HAS_N = 2 };
void *mprg_ingest_open_host(const char *paths, long long n_files, int n_threads);
/* the same on texts that are already in memory (text i = lens[i] bytes at texts[i]; not copied: keep them until _close) */
void *mprg_ingest_open_mem_host(const char *const *texts, const long long *lens, long long n_texts, int n_threads);
void mprg_ingest_info_host(void *handle, long long *info);
void mprg_ingest_fill_host(void *handle, uint8_t *arena, const long long *raw_off, char *titles, const long long *title_off,
                           int n_threads);
long long mprg_ingest_text_host(void *handle, long long i, const char **text);
void mprg_ingest_close_host(void *handle);
void mprg_encode_sizes_host(const char *prg, const long long *base, const long long *len, long long n, int n_threads,
                            int want_bin, int want_gfa, long long *bin_words, long long *gfa_bytes);
void mprg_encode_fill_host(const char *prg, const long long *base, const long long *len, long long n, int n_threads,
                           uint32_t *bin_out, const long long *bin_off, const long long *bin_words, char *gfa_out,
                           const long long *gfa_off, const long long *gfa_bytes, uint32_t *crc);
/* scatter-gather output: piece k = len[k] bytes at address addr[k] -> file offset file_off[k] of fd, n_threads threads; 0 or -1 */
int mprg_write_pieces_host(int fd, const long long *addr, const long long *len, const long long *file_off, long long n_pieces,
                           int n_threads);
/* CRC-32 of zip members given as pieces (member i = pieces first[i] .. first[i+1]-1; piece k = len[k] bytes at address addr[k]) */
void mprg_crc32_members_host(const long long *addr, const long long *len, const long long *first, long long n_members, int n_threads,
                             uint32_t *crc);

/* one-pass form of mprg_encode_sizes/fill_host: every locus encoded once into memory of an encode POOL (blocks owned by the
 * pool, reused after _reset); per locus the ADDRESS and size of its binary PRG / GFA text (-1 sizes: not covered by the one-pass
 * encoders) and the three CRCs.  The containers are written from address tables (mprg_write_pieces_host).  0, or -1: out of memory.
 * _info: {bytes mapped, bytes handed out since the last reset}. */
void *mprg_encode_pool_new_host(void);
void mprg_encode_pool_reset_host(void *pool);
void mprg_encode_pool_free_host(void *pool);
void mprg_encode_pool_info_host(void *pool, long long *info);
int mprg_encode_batch_host(void *pool, const char *prg, const long long *base, const long long *len, long long n, int n_threads,
                           int want_bin, int want_gfa, long long *bin_addr, long long *bin_words, long long *gfa_addr,
                           long long *gfa_bytes, uint32_t *crc);
/* running CRC-32 of one buffer, zlib.crc32(data, crc) (carry-less-multiplication folding where the host has PCLMULQDQ) */
uint32_t mprg_crc32_host(uint32_t crc, const void *data, long long len);

/* Device-memory, stream and event plumbing for hosts that bring no GPU framework of their own (a Go / C / Java host binding
 * this header; this repository's command line, which starts ~1 s earlier without importing torch).  Thin calls into the HIP
 * runtime the library is linked against.  Pointers: NULL on failure; ints: 0 or a negative code; mprg_last_error() says what.
 * mprg_rt_host_malloc: page-locked host memory that kernels may also write (result headers).  mprg_rt_memcpy_async kinds below;
 * pageable host memory makes a copy synchronous.  mprg_rt_event_query: 0 done, 1 not yet.  Streams are non-blocking streams;
 * every kernel entry point above takes one as its `stream` argument.  mprg_rt_init(device) makes `device` current for the CALLING
 * THREAD (HIP's current device is per thread; a new thread starts on device 0): call it in every thread that allocates or launches. */
enum { MPRG_RT_H2D = 1, MPRG_RT_D2H = 2, MPRG_RT_D2D = 3 };
int mprg_rt_device_count(void);
int mprg_rt_init(int device);
void *mprg_rt_malloc(long long nbytes);
int mprg_rt_free(void *p);
void *mprg_rt_host_malloc(long long nbytes);
int mprg_rt_host_free(void *p);
void *mprg_rt_stream_create(void);
int mprg_rt_stream_destroy(void *stream);
int mprg_rt_stream_sync(void *stream);
int mprg_rt_memcpy_async(void *dst, const void *src, long long nbytes, int kind, void *stream);
int mprg_rt_memset_async(void *dst, int value, long long nbytes, void *stream);
void *mprg_rt_event_create(int timing);
int mprg_rt_event_destroy(void *event);
int mprg_rt_event_record(void *event, void *stream);
int mprg_rt_event_sync(void *event);
int mprg_rt_event_query(void *event);
int mprg_rt_stream_wait_event(void *stream, void *event);
double mprg_rt_event_elapsed_ms(void *start, void *stop);

#ifdef __cplusplus
}
#endif
#endif
