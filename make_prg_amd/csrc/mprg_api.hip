// mprg_api.hip — C ABI (include/mprg.h) over the gfx950 kernels of the from_msa hot path.
// Build (product):  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared mprg_api.hip -o libmprg_hip.so
#include "mprg_platform.h"
#include "../../include/mprg.h"
#include <stdio.h>
#include <initializer_list>
#include <string.h>

#include "k_ingest.inc"
#include "k_columns.inc"
#include "k_partition.inc"
#include "k_rows.inc"
#include "k_kmer.inc"
#include "k_kmeans.inc"
#include "k_kmeans_lds.inc"
#include "k_cluster.inc"
#include "k_kloop.inc"
#include "k_emit.inc"
#include "k_forest.inc"
#include "host_encoders.inc"
#include "host_batch.inc"

static thread_local char g_err[512] = "";
static int fail(const char *what) { snprintf(g_err, sizeof g_err, "%s", what); return -1; }

static int check_launch(const char *name) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(g_err, sizeof g_err, "%s: %s", name, hipGetErrorString(e)); return -2; }
  return 0;
}

#include "runtime_api.inc"

#define BLOCK_VIEW 256
#include <stdlib.h>
// workgroup-size knobs of a few kernels (diagnostic runs): read ONCE, when the library is loaded
static int env_threads(const char *name, int dflt, int hi = 1024) {
  const char *e = getenv(name);
  const int v = e ? atoi(e) : dflt;
  return (v >= 64 && v <= hi && v % 64 == 0) ? v : dflt;
}
static const int g_pf_threads = env_threads("MPRG_PF_THREADS", BLOCK_VIEW);
static const int g_dd_threads = env_threads("MPRG_DD_THREADS", BLOCK_VIEW, 512);   // k_ungap_dedupe packs three 10-bit counters per scan
static const int g_km_threads = env_threads("MPRG_KM_THREADS", 256);
static const int g_km_wide_threads = env_threads("MPRG_KM_WIDE_THREADS", 1024);
static const int g_kp_threads = env_threads("MPRG_KP_THREADS", 0);
// the sample-sample tables of K6's global form by tiles of pairs through LDS (k_kmeans_prepare_tables_tiled); MPRG_KP_TILED=0: a thread per element
static const int g_kp_tiled = [] { const char *e = getenv("MPRG_KP_TILED"); return (e && atoi(e) == 0) ? 0 : 1; }();
// the selection of wide fits: predict() in a launch of its own, KPW_PARTS workgroups per fit (MPRG_KPW_SPLIT=0: inside the selection's one workgroup)
static const int g_kpw_split = [] { const char *e = getenv("MPRG_KPW_SPLIT"); return (e && atoi(e) == 0) ? 0 : 1; }();
static const int g_kms_threads = env_threads("MPRG_KMS_THREADS", 128, 128);          // the small KMeans form: 64 or 128 threads per fit
// the LDS form (k_kmeans_fit_lds): threads per fit by LDS class — the classes with fewer workgroups per CU get more threads
static const int g_kml_flags = [] { const char *e = getenv("MPRG_KML_FLAGS"); return e ? (atoi(e) & 0x7f) : 0; }();          // tuning switches (k_kmeans_lds.inc)
static const int g_kml_threads[KML_CLASSES] = {env_threads("MPRG_KML_THREADS0", 128, 256), env_threads("MPRG_KML_THREADS1", 128, 256),
                                               env_threads("MPRG_KML_THREADS2", 256, 1024), env_threads("MPRG_KML_THREADS3", 256, 1024),
                                               env_threads("MPRG_KML_THREADS4", 512, 1024), env_threads("MPRG_KML_THREADS5", 1024, 1024)};
// (the classes beyond 64 KB of LDS per workgroup — static + dynamic — need the limit raised once per kernel)
template <class K> static int kml_raise_lds(K kernel, bool *raised, const char *what) {
  if (*raised) return 0;
  if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KML_C5) != hipSuccess) return fail(what);
  *raised = true;
  return 0;
}
// small views by a wavefront each, several per workgroup (k_partition_wave, ...): MPRG_WAVE_VIEWS=0 keeps a workgroup per view
// mprg_cluster_further: problems that fit a workgroup's LDS in one workgroup and launch (k_cluster_further_one); MPRG_CF_ONE=0: two launches for all
static const int g_cf_one = [] { const char *e = getenv("MPRG_CF_ONE"); return (e && atoi(e) == 0) ? 0 : 1; }();
static const int g_pw_wave = [] { const char *e = getenv("MPRG_WAVE_VIEWS"); return (e && atoi(e) == 0) ? 0 : 1; }();

extern "C" {

#ifndef MPRG_BUILD_TAG
#define MPRG_BUILD_TAG "hip gfx950"
#endif
const char *mprg_version(void) { return "mprg 0.2 (" MPRG_BUILD_TAG ")"; }
const char *mprg_last_error(void) { return g_err; }

int mprg_device_cus(void) {
  int dev = 0; hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail("no HIP device");
  return p.multiProcessorCount;
}

int mprg_ingest(const uint8_t *raw, const int64_t *msa_table, int n_msas, int64_t n_tiles, const uint8_t *n_replacement,
                uint8_t *arena, int64_t arena_bytes, int32_t *status, void *stream) {
  if (n_msas <= 0) return 0;
  if (hipMemsetAsync(arena, 15, (size_t)arena_bytes, (hipStream_t)stream) != hipSuccess) return fail("memset");   // padding cells
  if (hipMemsetAsync(status, 0, sizeof(int32_t) * n_msas, (hipStream_t)stream) != hipSuccess) return fail("memset");
  if (n_tiles > 0) LAUNCH(k_ingest, n_tiles, 256, stream, raw, msa_table, n_msas, n_replacement, arena, status);
  return check_launch("k_ingest");
}

int mprg_column_residue_counts(const uint8_t *raw, const int64_t *table, const int32_t *work, int n_work, int32_t *out,
                               void *stream) {
  if (n_work <= 0) return 0;
  LAUNCH(k_column_residue_counts, n_work, 256, stream, raw, table, work, out);
  return check_launch("k_column_residue_counts");
}

// (the d_* forms: an entry point's launches with optional device-side counts — DsCount, mprg_platform.h; mprg_forest_level uses them)
static int d_column_masks(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int32_t *work,
                          int n_items, int rows_per_chunk, uint32_t *out_mask, void *stream, DsCount dc) {
  if (n_items <= 0) return 0;
  if (rows_per_chunk <= 0) return fail("rows_per_chunk must be positive");
  LAUNCH(k_column_masks, n_items, CM_THREADS, stream, arena, views, rowidx, work, rows_per_chunk, out_mask, dc);
  return check_launch("k_column_masks");
}
int mprg_column_masks(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int32_t *work,
                      int n_items, int rows_per_chunk, uint32_t *out_mask, void *stream) {
  return d_column_masks(arena, views, rowidx, work, n_items, rows_per_chunk, out_mask, stream, DS_HOST);
}

int mprg_compact_columns(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int32_t *work, int n_items,
                         int rows_per_chunk, const uint32_t *mask, uint8_t *out, const int64_t *out_off, int32_t *kept, void *stream) {
  if (n_items <= 0) return 0;
  if (rows_per_chunk <= 0) return fail("rows_per_chunk must be positive");
  LAUNCH(k_compact_columns, n_items, CC_THREADS, stream, arena, views, rowidx, work, rows_per_chunk, mask, out, out_off, kept);
  return check_launch("k_compact_columns");
}

static int d_partition(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views,
                       const uint32_t *mask, int min_match_length, const int32_t *work_rows, int n_work_rows,
                       uint32_t *maxrun, int32_t *stack, int32_t *ivflag, int32_t *iv, int32_t *n_iv, int32_t *status,
                       int32_t *view_out, int32_t *iv_packed, int32_t *iv_count, const int32_t *fused_list, int n_fused,
                       const int32_t *other_list, int n_other, void *stream, DsCount dc_views, DsCount dc_rows, DsCount dc_fused,
                       DsCount dc_other);
int mprg_partition(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views,
                   const uint32_t *mask, int min_match_length, const int32_t *work_rows, int n_work_rows,
                   uint32_t *maxrun, int32_t *stack, int32_t *ivflag, int32_t *iv, int32_t *n_iv, int32_t *status,
                   int32_t *view_out, int32_t *iv_packed, int32_t *iv_count, const int32_t *fused_list, int n_fused,
                   const int32_t *other_list, int n_other, void *stream) {
  return d_partition(arena, views, rowidx, n_views, mask, min_match_length, work_rows, n_work_rows, maxrun, stack, ivflag, iv, n_iv, status,
                     view_out, iv_packed, iv_count, fused_list, n_fused, other_list, n_other, stream, DS_HOST, DS_HOST, DS_HOST, DS_HOST);
}
static int d_partition(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views,
                       const uint32_t *mask, int min_match_length, const int32_t *work_rows, int n_work_rows,
                       uint32_t *maxrun, int32_t *stack, int32_t *ivflag, int32_t *iv, int32_t *n_iv, int32_t *status,
                       int32_t *view_out, int32_t *iv_packed, int32_t *iv_count, const int32_t *fused_list, int n_fused,
                       const int32_t *other_list, int n_other, void *stream, DsCount dc_views, DsCount dc_rows, DsCount dc_fused,
                       DsCount dc_other) {
  if (n_views <= 0) return 0;
  if (!fused_list && !other_list) { n_fused = 0; n_other = n_views; }
  else if (n_fused < 0 || n_other < 0 || (!dc_views.ds && n_fused + n_other != n_views)) return fail("mprg_partition: the two view lists must cover the views");
  else if ((n_fused > 0 && !fused_list) || (n_other > 0 && !other_list)) return fail("mprg_partition: a view list is missing");
  if (min_match_length < 1) return fail("mprg_partition: min_match_length must be positive");
  if (n_work_rows > 0) LAUNCH(k_gap_runs, n_work_rows, GR_ROWS, stream, arena, views, rowidx, work_rows, mask, maxrun, dc_rows);
  if (n_other > 0)
    LAUNCH(k_partition, n_other, BLOCK_VIEW, stream, other_list, arena, views, rowidx, mask, min_match_length, maxrun, stack,
           ivflag, iv, n_iv, status, view_out, dc_other);
  if (n_fused > 0) {
    // the fused list's SMALL views by a wavefront each (k_partition_wave: PW_WAVES per workgroup), the rest by a workgroup each:
    // both launches walk the same list, each leaves the other's views alone
    if (g_pw_wave) LAUNCH(k_partition_wave, (n_fused + PW_WAVES - 1) / PW_WAVES, PW_WAVES * WAVE, stream, fused_list, arena, views, rowidx,
                          min_match_length, iv, n_iv, status, view_out, n_fused, dc_fused);
    LAUNCH(k_partition_fused, n_fused, g_pf_threads, stream, fused_list, arena, views, rowidx, min_match_length, iv, n_iv, status,
           view_out, g_pw_wave, dc_fused);
  }
  if (view_out) {                                  // the packed list of all triples of the call
    if (!iv_packed || !iv_count) return fail("mprg_partition: view_out needs iv_packed and iv_count");
    LAUNCH(k_pack_scan, 1, 1024, stream, n_views, n_iv, view_out, iv_count, dc_views);
    LAUNCH(k_pack_copy, (n_views + PK_THREADS / WAVE - 1) / (PK_THREADS / WAVE), PK_THREADS, stream, n_views, views, iv, view_out,
           iv_packed, dc_views);
  }
  return check_launch("k_partition");
}

static int d_ungap_dedupe(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views, int kmer_size,
                          const int32_t *work_rows, int n_work_rows, uint8_t *ucodes, uint64_t *hashes, int32_t *ulen,
                          int32_t *rep_u, int32_t *rep_g, int32_t *d_of_row, int32_t *s_of_row, int32_t *reps_pos,
                          int32_t *reps_len, int32_t *seqrow, int64_t *occ_off, int64_t *summary, uint8_t *gcodes, void *stream,
                          DsCount dc_views, DsCount dc_rows, long long max_rows = 0) {
  if (n_views <= 0) return 0;
  // the SMALL views by a wavefront each, all stages in one launch (k_dedupe_wave); the launches below leave them alone
  if (g_pw_wave)
    LAUNCH(k_dedupe_wave, (n_views + DW_WAVES - 1) / DW_WAVES, DW_WAVES * WAVE, stream, arena, views, rowidx, n_views, kmer_size, ucodes, gcodes, hashes,
           ulen, rep_u, rep_g, d_of_row, s_of_row, reps_pos, reps_len, seqrow, occ_off, summary, dc_views);
  if (n_work_rows > 0) {
    LAUNCH(k_ungap_hash, n_work_rows, UG_ROWS, stream, arena, views, rowidx, work_rows, ucodes, hashes, ulen, gcodes, g_pw_wave, dc_rows);
    LAUNCH(k_ungap_hash_u, n_work_rows, UG_ROWS, stream, views, work_rows, (const uint8_t *)ucodes, hashes, (const int32_t *)ulen, g_pw_wave, dc_rows);
    // views of more rows than k_ungap_dedupe's LDS table holds: their row groups by a scan over (view, 256-row chunk) work items.
    // A list with one chunk per view has no such view; a device-counted list (capacities here) goes by the caller's bound on
    // the rows of a view (max_rows; 0: none known).
    if (max_rows > 0 ? max_rows > DD_ROWS : (dc_rows.ds != nullptr || n_work_rows > n_views))
      LAUNCH(k_dedupe_scan_big, n_work_rows, UG_ROWS, stream, arena, views, rowidx, work_rows, (const uint8_t *)ucodes, (const uint64_t *)hashes,
             (const int32_t *)ulen, rep_u, rep_g, dc_rows);
  }
  LAUNCH(k_ungap_dedupe, n_views, g_dd_threads, stream, arena, views, rowidx, kmer_size, ucodes, (const uint8_t *)gcodes, hashes, ulen, rep_u, rep_g,
         d_of_row, s_of_row, reps_pos, reps_len, seqrow, occ_off, summary, g_pw_wave, dc_views);
  return check_launch("k_ungap_dedupe");
}
int mprg_ungap_dedupe(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views, int kmer_size,
                      const int32_t *work_rows, int n_work_rows, uint8_t *ucodes, uint64_t *hashes, int32_t *ulen,
                      int32_t *rep_u, int32_t *rep_g, int32_t *d_of_row, int32_t *s_of_row, int32_t *reps_pos,
                      int32_t *reps_len, int32_t *seqrow, int64_t *occ_off, int64_t *summary, uint8_t *gcodes, void *stream) {
  return d_ungap_dedupe(arena, views, rowidx, n_views, kmer_size, work_rows, n_work_rows, ucodes, hashes, ulen, rep_u, rep_g, d_of_row, s_of_row,
                        reps_pos, reps_len, seqrow, occ_off, summary, gcodes, stream, DS_HOST, DS_HOST);
}

int mprg_kmer_dictionary(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size,
                         const uint8_t *ucodes, const int32_t *ulen, const int32_t *seqrow, int64_t *occ_off,
                         uint8_t *table, uint8_t *first_flag, int32_t *out_V, void *stream) {
  (void)ulen;
  if (n_probs <= 0) return 0;
  if (kmer_size < 1) return fail("k-mer size must be positive");
  LAUNCH(k_kmer_dictionary, n_probs, KD_THREADS, stream, views, prob, kmer_size, ucodes, seqrow, (const int64_t *)occ_off, table,
         first_flag, out_V, DS_HOST);
  return check_launch("k_kmer_dictionary");
}

int mprg_kmer_dictionary_parts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                               const int32_t *ulen, const int32_t *seqrow, int64_t *occ_off, uint8_t *table, uint8_t *first_flag,
                               int32_t *out_V, int parts, int32_t *part_counts, void *stream) {
  (void)ulen;
  if (n_probs <= 0) return 0;
  if (kmer_size < 1 || kmer_size > KMER_PACKED_MAX) return fail("mprg_kmer_dictionary_parts: k-mer sizes 1..16 (packed keys)");
  if (parts < 1 || parts > 1024 || !part_counts) return fail("mprg_kmer_dictionary_parts: parts must be 1..1024, part_counts given");
  LAUNCH2(k_kmer_dict_clear, n_probs, parts, 256, stream, views, prob, ucodes, seqrow, (const int64_t *)occ_off, table);
  LAUNCH2(k_kmer_dict_insert, n_probs, parts, 256, stream, views, prob, kmer_size, ucodes, seqrow, (const int64_t *)occ_off, table);
  LAUNCH2(k_kmer_dict_ids, n_probs, parts, 256, stream, views, prob, kmer_size, ucodes, seqrow, (const int64_t *)occ_off, table, first_flag,
          part_counts, out_V, 0);
  LAUNCH2(k_kmer_dict_ids, n_probs, parts, 256, stream, views, prob, kmer_size, ucodes, seqrow, (const int64_t *)occ_off, table, first_flag,
          part_counts, out_V, 1);
  return check_launch("k_kmer_dict_ids");
}
int mprg_kmer_counts_parts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                           const int32_t *ulen, const int32_t *seqrow, const int64_t *occ_off, const uint8_t *table,
                           double *xcounts, int parts, void *stream) {
  (void)ulen;
  if (n_probs <= 0) return 0;
  if (kmer_size < 1) return fail("k-mer size must be positive");
  if (parts < 1 || parts > 1024) return fail("mprg_kmer_counts_parts: parts must be 1..1024");
  LAUNCH2(k_kmer_counts, n_probs, parts, 512, stream, views, prob, kmer_size, ucodes, seqrow, occ_off, table, xcounts, DS_HOST);
  return check_launch("k_kmer_counts");
}
int mprg_kmer_counts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                     const int32_t *ulen, const int32_t *seqrow, const int64_t *occ_off, const uint8_t *table,
                     double *xcounts, void *stream) {
  return mprg_kmer_counts_parts(views, prob, n_probs, kmer_size, ucodes, ulen, seqrow, occ_off, table, xcounts, 1, stream);
}

int64_t mprg_kmeans_workspace_doubles(int64_t D, int64_t V, int k_max, int n_restart_slots) {
  if (k_max > KM_KMAX || V > (128LL << (KM_PW_DEPTH - 1))) return -1;       // V: depth of the pairwise-sum stack (k_kmeans.inc)
  return km_common_doubles_host(D, V) + (int64_t)n_restart_slots * km_restart_doubles_host(D, V);
}

static int d_kmeans_prepare(const int64_t *prob, int n_probs, const double *xcounts, double *ws, const int32_t *lds_list,
                            int n_lds, int64_t lds_bytes, const int32_t *other_list, int n_other, void *stream, DsCount dc);
int mprg_kmeans_prepare(const int64_t *prob, int n_probs, const double *xcounts, double *ws, const int32_t *lds_list,
                        int n_lds, int64_t lds_bytes, const int32_t *other_list, int n_other, void *stream) {
  return d_kmeans_prepare(prob, n_probs, xcounts, ws, lds_list, n_lds, lds_bytes, other_list, n_other, stream, DS_HOST);
}
int mprg_kmeans_prepare_big(const int64_t *prob, const double *xcounts, double *ws, const int32_t *list, int n_list, uint8_t *xbytes,
                            int with_tables, void *stream) {
  if (n_list <= 0) return 0;
  // the statistics by many workgroups per problem (k_kmeans_prepare_cols .. _norms: the values of k_kmeans_prepare's one workgroup)
  LAUNCH2(k_kmeans_prepare_cols, n_list, KPS_PARTS, 256, stream, list, prob, xcounts, ws);
  LAUNCH(k_kmeans_prepare_tol, n_list, 64, stream, list, prob, ws, with_tables ? 0 : 1, xbytes ? 1 : 0);
  LAUNCH2(k_kmeans_prepare_centre, n_list, KPS_PARTS, 256, stream, list, prob, xcounts, ws, xbytes);
  LAUNCH2(k_kmeans_prepare_norms, n_list, KPS_PARTS, 256, stream, list, prob, ws);
  if (with_tables) {          // a few big problems: more workgroups per problem than the pan-genome launches' KP_PARTS
    int parts = 4096 / n_list;
    parts = parts < KP_PARTS ? KP_PARTS : (parts > 1024 ? 1024 : parts);
    // by 32 x 32 tiles of sample pairs staged through LDS (with_tables = 2 or MPRG_KP_TILED=0: a thread per table element, the form before)
    if (with_tables != 2 && g_kp_tiled) LAUNCH(k_kmeans_prepare_tables_tiled, (long long)n_list * parts, KPT_THREADS, stream, list, prob, ws, parts, DS_HOST);
    else LAUNCH(k_kmeans_prepare_tables, (long long)n_list * parts, 256, stream, list, prob, xcounts, ws, xbytes, parts, DS_HOST);
  }
  return check_launch("k_kmeans_prepare");
}
// dc: device count of the ONE list the call holds (device-counted calls pass either lds_list or other_list)
static int d_kmeans_prepare(const int64_t *prob, int n_probs, const double *xcounts, double *ws, const int32_t *lds_list,
                            int n_lds, int64_t lds_bytes, const int32_t *other_list, int n_other, void *stream, DsCount dc) {
  if (n_probs <= 0) return 0;
  if (!lds_list && !other_list) { n_lds = 0; n_other = n_probs; }
  else if (n_lds + n_other != n_probs) return fail("mprg_kmeans_prepare: the two problem lists must cover the problems");
  if (n_lds > 0 && (lds_bytes <= 0 || lds_bytes > MPRG_KMEANS_PREPARE_LDS_MAX)) return fail("mprg_kmeans_prepare: lds_bytes out of range");
  if (n_other > 0) {
    LAUNCH(k_kmeans_prepare, n_other, 256, stream, other_list, prob, xcounts, ws, 0, (uint8_t *)nullptr, dc);
    if (g_kp_tiled) LAUNCH(k_kmeans_prepare_tables_tiled, (long long)n_other * KPT_PARTS, KPT_THREADS, stream, other_list, prob, ws, KPT_PARTS, dc);
    else LAUNCH(k_kmeans_prepare_tables, (long long)n_other * KP_PARTS, 256, stream, other_list, prob, xcounts, ws, (uint8_t *)nullptr, KP_PARTS, dc);
  }
  if (n_lds > 0) {
    if (lds_bytes > 64 * 1024) {     // beyond the default per-workgroup limit: gfx950 has 160 KB of LDS per CU, one such workgroup fits
      static bool raised = false;
      if (!raised) {
        if (hipFuncSetAttribute((const void *)k_kmeans_prepare_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                MPRG_KMEANS_PREPARE_LDS_MAX) != hipSuccess)
          return fail("mprg_kmeans_prepare: the device refuses MPRG_KMEANS_PREPARE_LDS_MAX bytes of LDS per workgroup");
        raised = true;
      }
    }
    // LDS decides how many of these workgroups a CU holds: the big classes get the threads the small ones get from residency
    const int prep_threads = g_kp_threads ? g_kp_threads : (lds_bytes > 64 * 1024 ? 1024 : (lds_bytes > 24 * 1024 ? 512 : 256));
    LAUNCH_LDS(k_kmeans_prepare_lds, n_lds, prep_threads, lds_bytes, stream, lds_list, prob, xcounts, ws, dc);
  }
  return check_launch("k_kmeans_prepare");
}

int mprg_kmeans_restarts(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *uniforms_dev,
                         const double *xcounts, double *ws, int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init > KM_RMAX) return fail("n_init must be <= 16");
  LAUNCH(k_kmeans_restart, n_fits, g_km_threads, stream, prob, kinfo, n_init, uniforms_dev, xcounts,
         ws, km_status);
  return check_launch("k_kmeans_restart");
}

int mprg_kmeans_fit(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init, const double *uniforms_dev,
                    const double *xcounts, double *ws, double *slot_ws, int64_t slot_stride_doubles, int n_slots,
                    int32_t *next_fit, int32_t *labels, double *km_info, int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init > KM_RMAX) return fail("n_init must be <= 16");
  static_assert(KM_SEL_LABELS * sizeof(int32_t) <= KM_XL_BYTES, "the selection's label staging lives in the restarts' LDS pool");
  if (!slot_ws) {                      // no scratch slots: one workgroup per fit on the restart regions of the problems' workspaces
    LAUNCH(k_kmeans_restart_select, n_fits, g_km_threads, stream, prob, kinfo, fit_list, n_init, uniforms_dev, xcounts,
           ws, labels, km_info, km_status);
    return check_launch("k_kmeans_restart_select");
  }
  if (fit_list) return fail("mprg_kmeans_fit: the persistent form takes no fit list");
  if (n_slots <= 0 || slot_stride_doubles <= 0) return fail("mprg_kmeans_fit: no scratch slots");
  if (hipMemsetAsync(next_fit, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return fail("memset");
  const int grid = n_fits < n_slots ? n_fits : n_slots;
  LAUNCH(k_kmeans_fit, grid, g_km_threads, stream, prob, kinfo, n_fits, n_init, uniforms_dev, xcounts, ws,
         slot_ws, (long long)slot_stride_doubles, next_fit, labels, km_info, km_status);
  return check_launch("k_kmeans_fit");
}

static int d_kmeans_fit_split(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init,
                              const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                              int32_t *km_status, int threads, const uint8_t *xbytes, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init < 1 || n_init > KM_RMAX) return fail("n_init must be 1..16");
  if (threads > 64)
    LAUNCH(k_kmeans_restart_wide, (long long)n_fits * n_init, threads, stream, prob, kinfo, fit_list, n_init, uniforms_dev, xcounts, ws, km_status,
           xbytes);
  else
    hipLaunchKernelGGL(k_kmeans_restart_one, dim3((unsigned)((long long)n_fits * n_init)), dim3(64), 0, (hipStream_t)stream, prob, kinfo, fit_list,
                       n_init, uniforms_dev, xcounts, ws, km_status);
  if (check_launch("k_kmeans_restart_one") != 0) return -2;
  // (the selection predicts the labels — D x k chains as long as the k-mer dictionary: a big fit's takes the wide workgroup and the bytes too)
  if (threads > 64 && g_kpw_split) {          // wide fits: predict() by KPW_PARTS workgroups per fit
    LAUNCH(k_kmeans_select_only_list, n_fits, threads, stream, prob, kinfo, fit_list, n_init, xcounts, ws, labels, km_info, xbytes);
    LAUNCH(k_kmeans_predict_list, (long long)n_fits * KPW_PARTS, threads, stream, prob, kinfo, fit_list, xcounts, ws, labels, km_info, xbytes, KPW_PARTS);
    LAUNCH(k_kmeans_predict_finish, (n_fits + 63) / 64, 64, stream, kinfo, fit_list, n_fits, km_info);
    return check_launch("k_kmeans_predict_list");
  }
  LAUNCH(k_kmeans_select_list, n_fits, threads > 64 ? threads : 256, stream, prob, kinfo, fit_list, n_init, xcounts, ws, labels, km_info, xbytes);
  return check_launch("k_kmeans_select_list");
}
int mprg_kmeans_fit_split(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init,
                          const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                          int32_t *km_status, void *stream) {
  return d_kmeans_fit_split(prob, kinfo, fit_list, n_fits, n_init, uniforms_dev, xcounts, ws, labels, km_info, km_status, 64, nullptr, stream);
}
int mprg_kmeans_fit_wide(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int n_init,
                         const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                         int32_t *km_status, const uint8_t *xbytes, void *stream) {
  return d_kmeans_fit_split(prob, kinfo, fit_list, n_fits, n_init, uniforms_dev, xcounts, ws, labels, km_info, km_status, g_km_wide_threads, xbytes,
                            stream);
}

int mprg_kmeans_wave_class(int64_t D, int64_t V, int k) { return (k < 2 || k > KM_KMAX) ? -1 : km_wave_class_host(D, V, k); }

int mprg_kmeans_fit_wave(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int lds_class, int n_init,
                         const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                         int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init > KM_RMAX) return fail("n_init must be <= 16");
#define KMW_LAUNCH(DBL) hipLaunchKernelGGL((k_kmeans_fit_wave<DBL>), dim3((unsigned)n_fits), dim3(64), 0, (hipStream_t)stream, prob, kinfo, \
                                           fit_list, n_init, uniforms_dev, xcounts, ws, labels, km_info, km_status)
  switch (lds_class) {
    case 0: KMW_LAUNCH(KMW_C0); break;
    case 1: KMW_LAUNCH(KMW_C1); break;
    case 2: KMW_LAUNCH(KMW_C2); break;
    case 3: KMW_LAUNCH(KMW_C3); break;
    default: return fail("mprg_kmeans_fit_wave: lds_class must be 0..3 (mprg_kmeans_wave_class)");
  }
#undef KMW_LAUNCH
  return check_launch("k_kmeans_fit_wave");
}

int mprg_argpartition(const double *values, int32_t *perm, int n, int kth, int32_t *ok, void *stream) {
  if (n <= 0 || kth < 0 || kth >= n) return fail("mprg_argpartition: kth out of range");
  LAUNCH(k_argpartition, 1, 64, stream, values, perm, n, kth, ok);
  return check_launch("k_argpartition");
}

int mprg_kmeans_small_class(int64_t D, int64_t V, int k, int n_init) {
  if (k < 2 || k > KM_KMAX || n_init > KMS_RMAX || n_init * D > KMS_LAB8) return -1;
  long long vp = (V + 3) & ~3LL; if (((vp >> 2) & 1) == 0) vp += 4;
  if (8 * ((V + 1) & ~1LL) + D * vp > KMS_POOL || (long long)n_init * 5 * D * 8 > KMS_POOL) return -1;
  return k <= 6 ? 0 : 1;
}

int mprg_kmeans_fit_small(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int small_class, int n_init,
                          const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                          int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init > KMS_RMAX) return fail("mprg_kmeans_fit_small: n_init must be <= 10");
#define KMS_LAUNCH(KCH) hipLaunchKernelGGL((k_kmeans_restart_select_small<KCH>), dim3((unsigned)n_fits), dim3(g_kms_threads), 0, (hipStream_t)stream, prob, \
                                           kinfo, fit_list, n_init, uniforms_dev, xcounts, ws, labels, km_info, km_status)
  if (small_class == 0) KMS_LAUNCH(36);
  else if (small_class == 1) KMS_LAUNCH(KM_KMAX * KM_KMAX);
  else return fail("mprg_kmeans_fit_small: small_class must be 0 or 1 (mprg_kmeans_small_class)");
#undef KMS_LAUNCH
  return check_launch("k_kmeans_restart_select_small");
}

int mprg_kmeans_lds_class(int64_t D, int64_t V, int k, int n_init) { return kml_class(D, V, k, n_init); }

int mprg_kmeans_fit_lds(const int64_t *prob, const int32_t *kinfo, const int32_t *fit_list, int n_fits, int lds_class, int n_init,
                        const double *uniforms_dev, const double *xcounts, double *ws, int32_t *labels, double *km_info,
                        int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init < 1 || n_init > KML_RMAX) return fail("mprg_kmeans_fit_lds: n_init must be 1..10");
  if (lds_class < 0 || lds_class >= KML_CLASSES) return fail("mprg_kmeans_fit_lds: lds_class must be 0..5 (mprg_kmeans_lds_class)");
  const int bytes = kml_class_bytes(lds_class);
  static bool raised = false;
  if (bytes > 56 * 1024 && kml_raise_lds(k_kmeans_fit_lds, &raised, "mprg_kmeans_fit_lds: the device refuses the LDS of the largest class") != 0) return -1;
  LAUNCH_LDS(k_kmeans_fit_lds, n_fits, g_kml_threads[lds_class], bytes, stream, prob, kinfo, fit_list, n_init, uniforms_dev, xcounts, ws, labels,
             km_info, km_status, bytes | (g_kml_flags << 24));
  return check_launch("k_kmeans_fit_lds");
}

int mprg_kmeans_select(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *xcounts,
                       double *ws, int32_t *labels, double *km_info, void *stream) {
  if (n_fits <= 0) return 0;
  LAUNCH(k_kmeans_select, n_fits, 256, stream, prob, kinfo, n_init, xcounts, ws, labels, km_info);
  return check_launch("k_kmeans_select");
}

static int d_cluster_further(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                             int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                             const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                             int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, const int32_t *kinfo,
                             void *stream, DsCount dc_cols, DsCount dc_rows, DsCount dc_probs, long long max_rows = 0);
int mprg_cluster_further(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                         int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                         const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                         int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, const int32_t *kinfo,
                         void *stream) {
  return d_cluster_further(arena, views, rowidx, prob, n_probs, k, d_of_row, labels, assign, work_cols, n_work_cols, work_rows, n_work_rows,
                           scratch, out_further, km_info, gcodes, kinfo, stream, DS_HOST, DS_HOST, DS_HOST);
}
int mprg_cluster_further_bounded(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                                 int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                                 const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                                 int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, const int32_t *kinfo,
                                 long long max_rows, void *stream) {
  return d_cluster_further(arena, views, rowidx, prob, n_probs, k, d_of_row, labels, assign, work_cols, n_work_cols, work_rows, n_work_rows,
                           scratch, out_further, km_info, gcodes, kinfo, stream, DS_HOST, DS_HOST, DS_HOST, max_rows);
}
static int d_cluster_further(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                             int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                             const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                             int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, const int32_t *kinfo,
                             void *stream, DsCount dc_cols, DsCount dc_rows, DsCount dc_probs, long long max_rows) {
  if (n_probs <= 0) return 0;
  if (k < 1 || k > KM_KMAX) return fail("mprg_cluster_further: k out of range");
  if (hipMemsetAsync(out_further, 0, sizeof(int32_t) * n_probs, (hipStream_t)stream) != hipSuccess) return fail("memset");
  // problems whose cells fit a workgroup's LDS: the whole check in ONE workgroup and launch (needs the dense gapped copy)
  const int one = (g_cf_one && gcodes) ? 1 : 0;
  if (one) LAUNCH(k_cluster_further_one, n_probs, CFO_THREADS, stream, views, prob, n_probs, k, d_of_row, labels, assign, km_info, out_further, gcodes,
                  kinfo, dc_probs);
  // problems of more than CF_ROWS rows: their majority strings by wide workgroups in a launch of their own over the same work items
  // (left out when the caller's bound on the rows of a view says there is none: max_rows, 0 = none known)
  const int big = (max_rows <= 0 || max_rows > CF_ROWS) ? 1 : 0;
  LAUNCH(k_cluster_majority, n_work_cols, CF_THREADS, stream, arena, views, rowidx, prob, work_cols, k, d_of_row, labels,
         assign, km_info, scratch, gcodes, kinfo, one, dc_cols);
  if (big) LAUNCH(k_cluster_majority_big, n_work_cols, CFB_THREADS, stream, arena, views, rowidx, prob, work_cols, k, d_of_row, labels,
                  scratch, gcodes, kinfo, dc_cols);
  LAUNCH(k_cluster_hamming, n_work_rows, CF_TILE, stream, arena, views, rowidx, prob, work_rows, d_of_row, labels,
         (const int32_t *)scratch, out_further, gcodes, kinfo, one ? k : 0, dc_rows);
  return check_launch("k_cluster_further");
}

static int d_cluster_loop(const int64_t *views, const int64_t *prob, int n_probs, int n_init, const double *uniforms_dev,
                          const int32_t *uniform_offsets_host, const double *xcounts, double *ws, const int32_t *d_of_row,
                          const uint8_t *gcodes, int32_t *scratch, int32_t *labels, int32_t *assign, double *km_info, int32_t *km_status,
                          int32_t *num_clusters, int32_t *active, int64_t *stats, int forms, void *stream, DsCount dc);
int mprg_cluster_loop(const int64_t *views, const int64_t *prob, int n_probs, int n_init, const double *uniforms_dev,
                      const int32_t *uniform_offsets_host, const double *xcounts, double *ws, const int32_t *d_of_row,
                      const uint8_t *gcodes, int32_t *scratch, int32_t *labels, int32_t *assign, double *km_info, int32_t *km_status,
                      int32_t *num_clusters, int32_t *active, int64_t *stats, int forms, void *stream) {
  return d_cluster_loop(views, prob, n_probs, n_init, uniforms_dev, uniform_offsets_host, xcounts, ws, d_of_row, gcodes, scratch, labels, assign,
                        km_info, km_status, num_clusters, active, stats, forms, stream, DS_HOST);
}
static int d_cluster_loop(const int64_t *views, const int64_t *prob, int n_probs, int n_init, const double *uniforms_dev,
                          const int32_t *uniform_offsets_host, const double *xcounts, double *ws, const int32_t *d_of_row,
                          const uint8_t *gcodes, int32_t *scratch, int32_t *labels, int32_t *assign, double *km_info, int32_t *km_status,
                          int32_t *num_clusters, int32_t *active, int64_t *stats, int forms, void *stream, DsCount dc) {
  if (n_probs <= 0) return 0;
  if (n_init < 1 || n_init > KM_RMAX) return fail("n_init must be 1..16");
  if (!gcodes || !uniform_offsets_host || !stats) return fail("mprg_cluster_loop: gcodes, uniform_offsets_host and stats are required");
  if (!(forms & (7 | MPRG_LOOP_LDS))) return fail("mprg_cluster_loop: forms must name a workgroup form (MPRG_LOOP_*)");
  KlUoff uoff;
  for (int k = 0; k <= KM_KMAX; ++k) uoff.v[k] = k >= 2 ? uniform_offsets_host[k] : 0;
  const bool small_ok = n_init <= KMS_RMAX;
  // the LDS form first, a launch per class in ascending order (a problem whose next round needs more LDS stays active for the next
  // launch; what no class holds is left to the general form below)
  static bool raised_loop = false;
  if ((forms & MPRG_LOOP_LDS) && n_init <= KML_RMAX &&
      kml_raise_lds(k_cluster_loop_lds, &raised_loop, "mprg_cluster_loop: the device refuses the LDS of the largest class") != 0) return -1;
  if ((forms & MPRG_LOOP_LDS) && n_init <= KML_RMAX)
    for (int c = 0; c < KML_CLASSES; ++c)
      LAUNCH_LDS(k_cluster_loop_lds, n_probs, g_kml_threads[c], kml_class_bytes(c), stream, prob, n_init, uniforms_dev, uoff, xcounts, ws, views, d_of_row,
                 gcodes, scratch, labels, assign, km_info, km_status, num_clusters, active, stats, c, kml_class_bytes(c) | (g_kml_flags << 24), dc);
  if (forms & MPRG_LOOP_GENERAL)
    LAUNCH(k_cluster_loop, n_probs, g_km_threads, stream, prob, n_init, uniforms_dev, uoff, xcounts, ws, views, d_of_row, gcodes, scratch, labels,
           assign, km_info, km_status, num_clusters, active, stats,
           (((forms & MPRG_LOOP_SKIP_SMALL) && small_ok) ? 1 : 0) | (((forms & MPRG_LOOP_SKIP_LDS) && n_init <= KML_RMAX) ? 2 : 0), dc);
#define KLS_LAUNCH(KCH, KHI) hipLaunchKernelGGL((k_cluster_loop_small<KCH, KHI>), dim3((unsigned)n_probs), dim3(128), 0, (hipStream_t)stream, prob, \
                                                 n_init, uniforms_dev, uoff, xcounts, ws, views, d_of_row, gcodes, scratch, labels, assign, \
                                                 km_info, km_status, num_clusters, active, stats, dc)
  if ((forms & MPRG_LOOP_SMALL_LOW) && small_ok) KLS_LAUNCH(36, 6);
  if ((forms & MPRG_LOOP_SMALL_HIGH) && small_ok) KLS_LAUNCH(KM_KMAX * KM_KMAX, KM_KMAX);
#undef KLS_LAUNCH
  return check_launch("k_cluster_loop");
}

int mprg_split_children(const int64_t *views, const int32_t *rowidx, const int64_t *prob, int n_probs,
                        const int64_t *split_info, const int32_t *d_of_row, const int32_t *s_of_row,
                        const int32_t *assign, int32_t *pool_out, int32_t *child_sizes, void *stream) {
  if (n_probs <= 0) return 0;
  LAUNCH(k_split_children, n_probs, 64, stream, views, rowidx, prob, split_info, d_of_row, s_of_row, assign, pool_out,
         child_sizes, DS_HOST);
  return check_launch("k_split_children");
}

int mprg_leaf_jobs(const int64_t *leaves, int64_t n_leaves, const int32_t *rowidx, const int32_t *reps_pos,
                   const int32_t *reps_len, int64_t *jobs, uint8_t *out, void *stream) {
  if (n_leaves <= 0) return 0;
  LAUNCH(k_leaf_jobs, (n_leaves + 255) / 256, 256, stream, leaves, (long long)n_leaves, rowidx, reps_pos, reps_len, jobs, out);
  return check_launch("k_leaf_jobs");
}

int mprg_emit_alleles(const uint8_t *arena, const int64_t *jobs, int64_t n_jobs, uint8_t *out, void *stream) {
  if (n_jobs <= 0) return 0;
  LAUNCH(k_emit_alleles, (n_jobs + 3) / 4, EMIT_THREADS, stream, arena, jobs, (long long)n_jobs, out);      // a wavefront per job
  return check_launch("k_emit_alleles");
}

// ---- the recursion forest on the device (k_forest.inc); F: host array of MPRG_F_FIELDS int64 ------------------------------
#define FP(T, f) ((T *)(uintptr_t)F[f])
#define FHDR FP(int64_t, MPRG_F_HDR)
static int kf_publish(const int64_t *F, void *stream, const char *name) {
  if (F[MPRG_F_HDR_HOST]) LAUNCH(k_hdr_publish, 1, 128, stream, (const int64_t *)FHDR, FP(int64_t, MPRG_F_HDR_HOST));
  return check_launch(name);
}
// Every step exists once, for both ways of running a level: `ds` == nullptr — the step's own entry point, called by a host that
// reads the totals back (hdr = MPRG_F_HDR, published to the host) — or the forest's device state — mprg_forest_level: the item
// counts in F are CAPACITIES, the exact counts are words of ds, the step's totals go to its block `hdr` inside ds.
// ck (optional, with ds): the capacity check of the step's totals, carried out by the launch that writes them (kf_scan)
struct KfStep { const int64_t *ds; int64_t *hdr; const KfCheck *ck = nullptr; };
static inline DsCount kf_dc(const KfStep &st, long long slot) { return DsCount{st.ds, (int)slot}; }
static inline int kf_slot(const KfStep &st, const int64_t *word) { return st.ds ? (int)(word - st.ds) : 0; }
static int kf_count_done(const int64_t *F, const KfStep &st, long long n, int m, DsCount dc, void *stream, const char *name) {
  if (kf_scan(FP(int64_t, MPRG_F_VALS), n, m, st.hdr, FP(int64_t, MPRG_F_SCAN_TMP), stream, dc, (int64_t *)st.ds, st.ck) != 0) return fail("scan");
  return st.ds ? check_launch(name) : kf_publish(F, stream, name);
}
static int kf_frontier_count(const int64_t *F, const KfStep &st, void *stream) {
  const long long n = F[MPRG_F_N];
  const DsCount dn = kf_dc(st, MPRG_DS_N);
  if (n > 0) LAUNCH(k_fr_count, KF_GRID(n), 256, stream, FP(const int64_t, MPRG_F_NODES), (long long)F[MPRG_F_F0], n, (int)F[MPRG_F_MIN_MATCH],
                    (int)F[MPRG_F_FUSED_ENABLED], FP(int64_t, MPRG_F_VALS), dn);
  return kf_count_done(F, st, n, 13, dn, stream, "k_fr_count");
}
static int kf_frontier_fill(const int64_t *F, const KfStep &st, void *stream) {
  const long long n = F[MPRG_F_N];
  if (n <= 0) return 0;
  LAUNCH(k_fr_fill, KF_GRID(n), 256, stream, FP(int64_t, MPRG_F_NODES), (long long)F[MPRG_F_F0], n, (int)F[MPRG_F_MIN_MATCH],
         (int)F[MPRG_F_FUSED_ENABLED], FP(const int64_t, MPRG_F_VALS), FP(const int64_t, MPRG_F_META), (int)F[MPRG_F_RPC_IDX],
         FP(int64_t, MPRG_F_VIEWS), FP(int64_t, MPRG_F_VIEW2NODE), FP(int32_t, MPRG_F_FUSED_LIST), FP(int32_t, MPRG_F_OTHER_LIST),
         FP(int32_t, MPRG_F_MASK_WORK), FP(int32_t, MPRG_F_GAP_WORK), kf_dc(st, MPRG_DS_N));
  return check_launch("k_fr_fill");
}
// frontier: the frontier step's block (device counts of the views); null in the host-counted form
static int kf_classify(const int64_t *F, const KfStep &st, const int64_t *frontier, void *stream) {
  const long long n = F[MPRG_F_N], nv = F[MPRG_F_N_VIEWS];
  const DsCount dn = kf_dc(st, MPRG_DS_N);
  if (nv > 0) LAUNCH(k_lv_status, KF_GRID(nv), 256, stream, FP(const int64_t, MPRG_F_NODES), (long long)F[MPRG_F_F0], nv,
                     FP(const int64_t, MPRG_F_VIEW2NODE), FP(const int32_t, MPRG_F_VIEW_OUT), FP(int32_t, MPRG_F_FAILED),
                     FP(int64_t, MPRG_F_ERR_FIRST), kf_dc(st, kf_slot(st, frontier)));
  if (n > 0) LAUNCH(k_lv_classify, KF_GRID(n), 256, stream, FP(int64_t, MPRG_F_NODES), (long long)F[MPRG_F_F0], n, (int)F[MPRG_F_LVL],
                    FP(const int32_t, MPRG_F_VIEW_OUT), FP(const int32_t, MPRG_F_FAILED), FP(int64_t, MPRG_F_VALS), dn);
  return kf_count_done(F, st, n, 7, dn, stream, "k_lv_classify");
}
static int kf_children(const int64_t *F, const KfStep &st, void *stream) {
  const long long n = F[MPRG_F_N];
  if (n <= 0) return 0;
  const DsCount dn = kf_dc(st, MPRG_DS_N);
  LAUNCH(k_lv_children, (n + 3) / 4, 256, stream, FP(int64_t, MPRG_F_NODES), (long long)F[MPRG_F_F0], n, (long long)F[MPRG_F_N_NODES],
         FP(const int64_t, MPRG_F_VALS), FP(const int32_t, MPRG_F_VIEW_OUT), FP(const int32_t, MPRG_F_IV_PACKED), dn);
  if (F[MPRG_F_NSEL] > 0)
    LAUNCH(k_lv_sub, KF_GRID(n), 256, stream, FP(const int64_t, MPRG_F_NODES), (long long)F[MPRG_F_F0], n, FP(const int64_t, MPRG_F_VALS),
           FP(const int64_t, MPRG_F_VIEWS), FP(int64_t, MPRG_F_SUB), FP(int64_t, MPRG_F_SELNODE), FP(int32_t, MPRG_F_DD_WORK), dn);
  return check_launch("k_lv_children");
}
// n_sel: device count of the selected views (a word of the classify step's block)
static int kf_cluster_count(const int64_t *F, const KfStep &st, const int64_t *n_sel, void *stream) {
  const long long n = F[MPRG_F_NSEL];
  const DsCount dn = kf_dc(st, kf_slot(st, n_sel));
  if (n > 0) LAUNCH(k_cl_classify, KF_GRID(n), 256, stream, FP(int64_t, MPRG_F_NODES), n, FP(const int64_t, MPRG_F_SELNODE),
                    FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_SUMMARY), (int)F[MPRG_F_MAX_NESTING], FP(int64_t, MPRG_F_VALS), dn);
  return kf_count_done(F, st, n, 5, dn, stream, "k_cl_classify");
}
static int kf_cluster_fill(const int64_t *F, const KfStep &st, const int64_t *n_sel, void *stream) {
  const long long n = F[MPRG_F_NSEL];
  if (n <= 0 || F[MPRG_F_NPQ] <= 0) return 0;
  LAUNCH(k_cl_fill_pq, KF_GRID(n), 256, stream, FP(const int64_t, MPRG_F_NODES), n, FP(const int64_t, MPRG_F_SELNODE),
         FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_SUMMARY), FP(const int64_t, MPRG_F_VALS), FP(int64_t, MPRG_F_T1),
         FP(int32_t, MPRG_F_WORK_COLS), FP(int32_t, MPRG_F_WORK_ROWS), kf_dc(st, kf_slot(st, n_sel)));
  return check_launch("k_cl_fill_pq");
}
static int kf_problems_count(const int64_t *F, const KfStep &st, const int64_t *n_pq, void *stream) {
  const long long n = F[MPRG_F_NPQ];
  const DsCount dn = kf_dc(st, kf_slot(st, n_pq));
  if (n > 0) LAUNCH(k_pr_count, KF_GRID(n), 256, stream, n, FP(const int64_t, MPRG_F_T1), FP(const int32_t, MPRG_F_FURTHER),
                    FP(const int64_t, MPRG_F_SUMMARY), FP(int64_t, MPRG_F_VALS), dn);
  return kf_count_done(F, st, n, 4, dn, stream, "k_pr_count");
}
static int kf_problems_fill(const int64_t *F, const KfStep &st, const int64_t *n_pq, void *stream) {
  const long long n = F[MPRG_F_NPQ];
  if (n <= 0 || F[MPRG_F_P] <= 0) return 0;
  LAUNCH(k_pr_fill, KF_GRID(n), 256, stream, n, FP(const int64_t, MPRG_F_T1), FP(const int32_t, MPRG_F_FURTHER),
         FP(const int64_t, MPRG_F_SUMMARY), FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_VALS), FP(int64_t, MPRG_F_PTAB0),
         kf_dc(st, kf_slot(st, n_pq)));
  return check_launch("k_pr_fill");
}
// n_p: device count of the problems (a word of the problems step's block)
static int kf_sizes_count(const int64_t *F, const KfStep &st, const int64_t *n_p, void *stream) {
  const long long P = F[MPRG_F_P];
  const DsCount dn = kf_dc(st, kf_slot(st, n_p));
  // (the header collects atomics; a block of the device state is zero already: mprg_forest_state_init / _rewind)
  if (!st.ds && hipMemsetAsync(st.hdr, 0, sizeof(int64_t) * MPRG_FOREST_HDR, (hipStream_t)stream) != hipSuccess) return fail("memset");
  if (P > 0) LAUNCH(k_sz_count, KF_GRID(P), 256, stream, P, FP(const int64_t, MPRG_F_PTAB0), FP(const int32_t, MPRG_F_DV),
                    FP(const int64_t, MPRG_F_SUB), (int)F[MPRG_F_N_INIT], FP(int64_t, MPRG_F_VALS), st.hdr, dn);
  return kf_count_done(F, st, P, 9, dn, stream, "k_sz_count");
}
static int kf_sizes_fill(const int64_t *F, const KfStep &st, const int64_t *n_p, void *stream) {
  const long long P = F[MPRG_F_P];
  if (P <= 0) return 0;
  LAUNCH(k_sz_fill, KF_GRID(P), 256, stream, P, FP(const int64_t, MPRG_F_PTAB0), FP(const int32_t, MPRG_F_DV), FP(const int64_t, MPRG_F_SUB),
         FP(const int64_t, MPRG_F_VALS), st.hdr, FP(int64_t, MPRG_F_PTAB), FP(int32_t, MPRG_F_CLS_LISTS), FP(int32_t, MPRG_F_NUM_CLUSTERS),
         FP(int32_t, MPRG_F_ACTIVE), FP(int32_t, MPRG_F_WORK_COLS), FP(int32_t, MPRG_F_WORK_ROWS), kf_dc(st, kf_slot(st, n_p)));
  return check_launch("k_sz_fill");
}
static int kf_splits_count(const int64_t *F, const KfStep &st, const int64_t *n_p, void *stream) {
  const long long P = F[MPRG_F_P];
  const DsCount dn = kf_dc(st, kf_slot(st, n_p));
  if (P > 0) LAUNCH(k_sp_count, KF_GRID(P), 256, stream, P, FP(const int64_t, MPRG_F_PTAB), FP(const int32_t, MPRG_F_NUM_CLUSTERS),
                    FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_SUMMARY), FP(int64_t, MPRG_F_VALS), dn);
  return kf_count_done(F, st, P, 3, dn, stream, "k_sp_count");
}
static int kf_splits_fill(const int64_t *F, const KfStep &st, const int64_t *n_p, void *stream) {
  const long long P = F[MPRG_F_P];
  if (P <= 0 || F[MPRG_F_NSPLITS] <= 0) return 0;
  LAUNCH(k_sp_fill, KF_GRID(P), 256, stream, P, FP(const int64_t, MPRG_F_PTAB), FP(const int32_t, MPRG_F_NUM_CLUSTERS),
         FP(const int64_t, MPRG_F_SELNODE), FP(const int64_t, MPRG_F_VALS), (long long)F[MPRG_F_POOL_USED], FP(int64_t, MPRG_F_SPT),
         FP(int64_t, MPRG_F_SP), FP(int64_t, MPRG_F_SPLITNODE), kf_dc(st, kf_slot(st, n_p)));
  return check_launch("k_sp_fill");
}
static int kf_split_children(const int64_t *F, const KfStep &st, const int64_t *n_splits, void *stream) {
  const long long ns = F[MPRG_F_NSPLITS];
  if (ns <= 0) return 0;
  LAUNCH(k_sp_children, (ns + 3) / 4, 256, stream, FP(int64_t, MPRG_F_NODES), ns, (long long)F[MPRG_F_N_NODES], FP(const int64_t, MPRG_F_SP),
         FP(const int64_t, MPRG_F_SPLITNODE), FP(const int32_t, MPRG_F_CHILD_SIZES), FP(const int64_t, MPRG_F_SPT),
         FP(const int64_t, MPRG_F_SUMMARY), kf_dc(st, kf_slot(st, n_splits)));
  return check_launch("k_sp_children");
}
#define KF_HOST KfStep{nullptr, FHDR}
int mprg_forest_frontier_count(const int64_t *F, void *stream) { return kf_frontier_count(F, KF_HOST, stream); }
int mprg_forest_frontier_fill(const int64_t *F, void *stream) { return kf_frontier_fill(F, KF_HOST, stream); }
int mprg_forest_classify(const int64_t *F, void *stream) { return kf_classify(F, KF_HOST, nullptr, stream); }
int mprg_forest_children(const int64_t *F, void *stream) { return kf_children(F, KF_HOST, stream); }
int mprg_forest_cluster_count(const int64_t *F, void *stream) { return kf_cluster_count(F, KF_HOST, nullptr, stream); }
int mprg_forest_cluster_fill(const int64_t *F, void *stream) { return kf_cluster_fill(F, KF_HOST, nullptr, stream); }
int mprg_forest_problems_count(const int64_t *F, void *stream) { return kf_problems_count(F, KF_HOST, nullptr, stream); }
int mprg_forest_problems_fill(const int64_t *F, void *stream) { return kf_problems_fill(F, KF_HOST, nullptr, stream); }
int mprg_forest_sizes_count(const int64_t *F, void *stream) { return kf_sizes_count(F, KF_HOST, nullptr, stream); }
int mprg_forest_sizes_fill(const int64_t *F, void *stream) { return kf_sizes_fill(F, KF_HOST, nullptr, stream); }
int mprg_forest_kloop_advance(const int64_t *F, int k, void *stream) {
  const long long P = F[MPRG_F_P];
  if (P <= 0) return 0;
  if (k < 2 || k > KM_KMAX + 1) return fail("mprg_forest_kloop_advance: k out of range");
  if (hipMemsetAsync(FHDR + 83, 0, sizeof(int64_t), (hipStream_t)stream) != hipSuccess) return fail("memset");
  if (hipMemsetAsync(FHDR + 86, 0, KL_LISTS * sizeof(int64_t), (hipStream_t)stream) != hipSuccess) return fail("memset");
  LAUNCH(k_kl_advance, KF_GRID(P), 256, stream, P, k, (int)F[MPRG_F_N_INIT], (int)F[MPRG_F_KM_MODE], FP(const int64_t, MPRG_F_PTAB), FP(const int64_t, MPRG_F_SUB),
         FP(int32_t, MPRG_F_NUM_CLUSTERS), FP(int32_t, MPRG_F_ACTIVE), FP(int32_t, MPRG_F_KINFO), FP(const double, MPRG_F_KM_INFO), FP(int32_t, MPRG_F_KM_STATUS),
         FP(const int32_t, MPRG_F_FURTHER), (int)F[MPRG_F_UOFF + (k <= KM_KMAX ? k : 0)], FP(int32_t, MPRG_F_FIT_LISTS), FHDR, FP(const double, MPRG_F_WS));
  return kf_publish(F, stream, "k_kl_advance");
}
int mprg_kmeans_speculative_kinfo(const int64_t *prob, int n_probs, int n_init, int mode, const int32_t *uniform_offsets_host, long long labels_per_k,
                                  int32_t *kinfo_out, const double *ws, void *stream) {
  if (n_probs <= 0) return 0;
  if (!uniform_offsets_host || !kinfo_out) return fail("mprg_kmeans_speculative_kinfo: uniform offsets / output missing");
  KlUoffs u;
  for (int k = 0; k <= KM_KMAX; ++k) u.v[k] = uniform_offsets_host[k];
  LAUNCH(k_kl_speculate, KF_GRID((long long)n_probs * (KM_KMAX - 1)), 256, stream, (long long)n_probs, n_init, mode, prob, u, labels_per_k, kinfo_out, ws);
  return check_launch("k_kl_speculate");
}
int mprg_forest_splits_count(const int64_t *F, void *stream) { return kf_splits_count(F, KF_HOST, nullptr, stream); }
int mprg_forest_splits_fill(const int64_t *F, void *stream) { return kf_splits_fill(F, KF_HOST, nullptr, stream); }
int mprg_forest_split_children(const int64_t *F, void *stream) { return kf_split_children(F, KF_HOST, nullptr, stream); }

int mprg_forest_state_init(int64_t *ds, long long n_words, long long n_roots, void *stream) {
  if (n_words < MPRG_DS_GLOBAL) return fail("mprg_forest_state_init: the device state is at least MPRG_DS_GLOBAL words");
  if (hipMemsetAsync(ds, 0, sizeof(int64_t) * n_words, (hipStream_t)stream) != hipSuccess) return fail("memset");
  LAUNCH(k_ds_init, 1, 64, stream, ds, n_roots);
  return check_launch("k_ds_init");
}
int mprg_forest_state_rewind(int64_t *ds, long long n_words, long long level, void *stream) {
  if (!ds || level < 0 || MPRG_DS_GLOBAL + (level + 1) * MPRG_DS_LEVEL_WORDS > n_words)
    return fail("mprg_forest_state_rewind: no device state / a level outside it");
  LAUNCH(k_ds_rewind, 1, 64, stream, ds, level, n_words);
  return check_launch("k_ds_rewind");
}
// One recursion level, every step enqueued, nothing read back (include/mprg.h: "a recursion level WITHOUT a host wait").
int mprg_forest_level(const int64_t *F, void *stream) {
  int64_t *ds = FP(int64_t, MPRG_F_DS);
  if (!ds) return fail("mprg_forest_level: MPRG_F_DS is not set");
  const long long L = F[MPRG_F_LEVEL_INDEX];
  const int64_t *C = F + MPRG_F_CAP;
  auto blk = [&](int s) { return ds + MPRG_DS_GLOBAL + (L * 6 + s) * MPRG_FOREST_HDR; };
  auto slot = [&](const int64_t *w) { return DsCount{ds, (int)(w - ds)}; };
  const long long BIG = 0x7fffffffffffffffLL;
  // caps: up to 16 column capacities (BIG: not limited), then two optional sum checks — checked by the step's scan (kf_scan)
  auto check = [&](int step, int m, std::initializer_list<long long> caps, int col_a = -1, int field_a = 0, long long cap_a = 0, int col_b = -1,
                   int field_b = 0, long long cap_b = 0) {
    KfCheck ck;
    int q = 0;
    for (long long v : caps) ck.caps.cap[q++] = v;
    for (; q < 16; ++q) ck.caps.cap[q] = BIG;
    ck.m = m; ck.code = 100 * (L + 1) + step;
    ck.col_a = col_a; ck.field_a = field_a; ck.cap_a = cap_a; ck.col_b = col_b; ck.field_b = field_b; ck.cap_b = cap_b;
    return ck;
  };
  const uint8_t *arena = FP(const uint8_t, MPRG_F_ARENA);
  const int32_t *pool = FP(const int32_t, MPRG_F_POOL);
  const int Lm = (int)F[MPRG_F_MIN_MATCH];
  // ---- S1 frontier -> views -> column masks, partition
  const long long cap_n = F[MPRG_F_N], cap_na = F[MPRG_F_N_VIEWS];
  LAUNCH(k_ds_begin, 1, 64, stream, ds, blk(MPRG_STEP_FRONTIER), cap_n, 100 * (L + 1) + MPRG_STEP_BEGIN);
  if (cap_n <= 0) {                              // the plan ends here: a frontier that is not empty is an overflow (k_ds_begin)
    LAUNCH(k_ds_advance, 1, 64, stream, ds, (const int64_t *)blk(MPRG_STEP_CLASSIFY), (const int64_t *)blk(MPRG_STEP_SPLITS));
    return check_launch("k_ds_advance");
  }
  int64_t *b0 = blk(MPRG_STEP_FRONTIER);
  long long item_caps[5] = {BIG, BIG, BIG, BIG, BIG};
  item_caps[(int)F[MPRG_F_RPC_IDX]] = C[MPRG_CAP_ITEMS];
  const KfCheck ck0 = check(MPRG_STEP_FRONTIER, 11, {cap_na, C[MPRG_CAP_TCOLS], BIG, C[MPRG_CAP_NFUSED], C[MPRG_CAP_NOTHER], item_caps[0], item_caps[1],
                                                     item_caps[2], item_caps[3], item_caps[4], C[MPRG_CAP_NGAP]});
  const KfStep st0{ds, b0, &ck0};
  if (kf_frontier_count(F, st0, stream) != 0) return -1;
  if (cap_na > 0) {
    if (kf_frontier_fill(F, st0, stream) != 0) return -1;
    const int rpc = (int)F[MPRG_F_RPC_IDX];
    if (C[MPRG_CAP_NOTHER] > 0 &&
        d_column_masks(arena, FP(const int64_t, MPRG_F_VIEWS), pool, FP(const int32_t, MPRG_F_MASK_WORK), (int)C[MPRG_CAP_ITEMS], 1024 >> rpc,
                       FP(uint32_t, MPRG_F_MASK), stream, slot(b0 + 5 + rpc)) != 0) return -1;
    const bool lists = C[MPRG_CAP_NFUSED] > 0;
    if (d_partition(arena, FP(const int64_t, MPRG_F_VIEWS), pool, (int)cap_na, FP(const uint32_t, MPRG_F_MASK), Lm, FP(const int32_t, MPRG_F_GAP_WORK),
                    (int)C[MPRG_CAP_NGAP], FP(uint32_t, MPRG_F_MAXRUN), FP(int32_t, MPRG_F_STACK), FP(int32_t, MPRG_F_IVFLAG), FP(int32_t, MPRG_F_IV),
                    FP(int32_t, MPRG_F_NIV), FP(int32_t, MPRG_F_STATUS), FP(int32_t, MPRG_F_VIEW_OUT), FP(int32_t, MPRG_F_IV_PACKED),
                    FP(int32_t, MPRG_F_IVC), lists ? FP(const int32_t, MPRG_F_FUSED_LIST) : nullptr, lists ? (int)C[MPRG_CAP_NFUSED] : 0,
                    lists ? FP(const int32_t, MPRG_F_OTHER_LIST) : nullptr, lists ? (int)C[MPRG_CAP_NOTHER] : 0, stream, slot(b0 + 0), slot(b0 + 10),
                    slot(b0 + 3), lists ? slot(b0 + 4) : slot(b0 + 0)) != 0) return -1;
  }
  // ---- S2 classify, children of multi-interval nodes, the selected views
  int64_t *b1 = blk(MPRG_STEP_CLASSIFY);
  const long long cap_sel = F[MPRG_F_NSEL];
  const KfCheck ck1 = check(MPRG_STEP_CLASSIFY, 6, {BIG, cap_sel, C[MPRG_CAP_SROWS], C[MPRG_CAP_UBYTES], C[MPRG_CAP_SCOLS], C[MPRG_CAP_NDD]}, 0,
                            MPRG_DS_NNODES, C[MPRG_CAP_NODES]);
  const KfStep st1{ds, b1, &ck1};
  if (kf_classify(F, st1, b0 + 0, stream) != 0) return -1;
  if (kf_children(F, st1, stream) != 0) return -1;
  LAUNCH(k_ds_add, 1, 64, stream, ds, (int)MPRG_DS_NNODES, (const int64_t *)b1);
  int64_t *b5 = blk(MPRG_STEP_SPLITS);
  if (cap_sel > 0) {
    // ---- row groups of the selected views; S3 which candidates go on
    if (d_ungap_dedupe(arena, FP(const int64_t, MPRG_F_SUB), pool, (int)cap_sel, Lm, FP(const int32_t, MPRG_F_DD_WORK), (int)C[MPRG_CAP_NDD],
                       FP(uint8_t, MPRG_F_UCODES), FP(uint64_t, MPRG_F_HASHES), FP(int32_t, MPRG_F_ULEN), FP(int32_t, MPRG_F_REP_U),
                       FP(int32_t, MPRG_F_REP_G), FP(int32_t, MPRG_F_D_OF_ROW), FP(int32_t, MPRG_F_S_OF_ROW), FP(int32_t, MPRG_F_REPS_POS),
                       FP(int32_t, MPRG_F_REPS_LEN), FP(int32_t, MPRG_F_SEQROW), FP(int64_t, MPRG_F_OCC_OFF), FP(int64_t, MPRG_F_SUMMARY),
                       FP(uint8_t, MPRG_F_GCODES), stream, slot(b1 + 1), slot(b1 + 5), F[MPRG_F_MAX_ROWS]) != 0) return -1;
    int64_t *b2 = blk(MPRG_STEP_CLUSTER);
    const long long cap_pq = F[MPRG_F_NPQ];
    const KfCheck ck2 = check(MPRG_STEP_CLUSTER, 3, {cap_pq, C[MPRG_CAP_WC], C[MPRG_CAP_WR]});
    const KfStep st2{ds, b2, &ck2};
    if (kf_cluster_count(F, st2, b1 + 1, stream) != 0) return -1;
    if (cap_pq > 0) {
      if (kf_cluster_fill(F, st2, b1 + 1, stream) != 0) return -1;
      // cluster_sequences.py:256: `while cluster_further(...)` is evaluated before any KMeans (k = 1, one cluster)
      if (d_cluster_further(arena, FP(const int64_t, MPRG_F_SUB), pool, FP(const int64_t, MPRG_F_T1), (int)cap_pq, 1, FP(const int32_t, MPRG_F_D_OF_ROW),
                            nullptr, nullptr, FP(const int32_t, MPRG_F_WORK_COLS), (int)C[MPRG_CAP_WC], FP(const int32_t, MPRG_F_WORK_ROWS),
                            (int)C[MPRG_CAP_WR], FP(int32_t, MPRG_F_CF_SCRATCH), FP(int32_t, MPRG_F_FURTHER), nullptr, FP(const uint8_t, MPRG_F_GCODES),
                            nullptr, stream, slot(b2 + 1), slot(b2 + 2), slot(b2 + 0), F[MPRG_F_MAX_ROWS]) != 0) return -1;
      // ---- S4 the clustering problems, their k-mer dictionaries
      int64_t *b3 = blk(MPRG_STEP_PROBLEMS);
      const long long cap_p = F[MPRG_F_P];
      const KfCheck ck3 = check(MPRG_STEP_PROBLEMS, 4, {cap_p, C[MPRG_CAP_TABLE], C[MPRG_CAP_FLAG], C[MPRG_CAP_LO]});
      const KfStep st3{ds, b3, &ck3};
      if (kf_problems_count(F, st3, b2 + 0, stream) != 0) return -1;
      if (cap_p > 0) {
        if (kf_problems_fill(F, st3, b2 + 0, stream) != 0) return -1;
        const DsCount dp = slot(b3 + 0);
        LAUNCH(k_kmer_dictionary, cap_p, KD_THREADS, stream, FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_PTAB0), Lm,
               FP(const uint8_t, MPRG_F_UCODES), FP(const int32_t, MPRG_F_SEQROW), FP(const int64_t, MPRG_F_OCC_OFF), FP(uint8_t, MPRG_F_TABLE),
               FP(uint8_t, MPRG_F_FLAG), FP(int32_t, MPRG_F_DV), dp);
        // ---- S5 count matrices, workspaces, launch classes
        int64_t *b4 = blk(MPRG_STEP_SIZES);
        const KfCheck ck4 = check(MPRG_STEP_SIZES, 7, {C[MPRG_CAP_XD], C[MPRG_CAP_WSD], C[MPRG_CAP_CLS], C[MPRG_CAP_CLS + 1], C[MPRG_CAP_CLS + 2],
                                                       C[MPRG_CAP_CLS + 3], C[MPRG_CAP_CLS + 4]});
        const KfStep st4{ds, b4, &ck4};
        if (kf_sizes_count(F, st4, b3 + 0, stream) != 0) return -1;
        // (b4[16 + c]: the largest LDS need of class c; class 4 = the global form, whose "need" is the size of the problem's count matrix:
        //  MPRG_CAP_BIG leaves a level that holds a BIG problem to the host, which has wider kernels for it; code: step 7)
        {
          DsCaps lcaps;
          for (int q = 0; q < 16; ++q) lcaps.cap[q] = BIG;
          for (int q = 0; q < 4; ++q) lcaps.cap[q] = C[MPRG_CAP_LDS + q];
          if (C[MPRG_CAP_BIG] > 0) for (int q = 0; q < 5; ++q) if (lcaps.cap[q] > C[MPRG_CAP_BIG]) lcaps.cap[q] = C[MPRG_CAP_BIG];
          LAUNCH(k_ds_check, 1, 64, stream, ds, (const int64_t *)(b4 + 16), 5, lcaps, 100 * (L + 1) + MPRG_STEP_SIZES_SHAPE, -1, 0, 0LL, -1, 0, 0LL);
        }
        if (kf_sizes_fill(F, st4, b3 + 0, stream) != 0) return -1;
        LAUNCH(k_kmer_counts, cap_p, 512, stream, FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_PTAB), Lm, FP(const uint8_t, MPRG_F_UCODES),
               FP(const int32_t, MPRG_F_SEQROW), FP(const int64_t, MPRG_F_OCC_OFF), FP(const uint8_t, MPRG_F_TABLE), FP(double, MPRG_F_X), dp);
        for (int c = 0; c < 5; ++c) {
          const long long n_c = C[MPRG_CAP_CLS + c];
          if (n_c <= 0) continue;
          const int32_t *lst = FP(const int32_t, MPRG_F_CLS_LISTS) + c * cap_p;
          const int rc = c < 4 ? d_kmeans_prepare(FP(const int64_t, MPRG_F_PTAB), (int)n_c, FP(const double, MPRG_F_X), FP(double, MPRG_F_WS), lst, (int)n_c,
                                                  C[MPRG_CAP_LDS + c], nullptr, 0, stream, slot(b4 + 2 + c))
                               : d_kmeans_prepare(FP(const int64_t, MPRG_F_PTAB), (int)n_c, FP(const double, MPRG_F_X), FP(double, MPRG_F_WS), nullptr, 0, 0,
                                                  lst, (int)n_c, stream, slot(b4 + 2 + c));
          if (rc != 0) return -1;
        }
        // ---- S6 the clustering loop of every problem (cluster_sequences.py:256-274), statistics into the sizes block.  The general
        //      form and the small forms are independent launches: with a side stream (MPRG_F_SIDE_STREAM) they run side by side —
        //      one form's last, longest problems overlap the other's work
        {
          const int forms = (int)F[MPRG_F_LOOP_FORMS];
          void *side = (void *)(uintptr_t)F[MPRG_F_SIDE_STREAM];
          const bool lds = (forms & MPRG_LOOP_LDS) != 0;
          const bool split = side && (forms & MPRG_LOOP_GENERAL) && (lds || (forms & (MPRG_LOOP_SMALL_LOW | MPRG_LOOP_SMALL_HIGH)));
          hipEvent_t e_fork = nullptr, e_join = nullptr;
          if (split) {
            if (hipEventCreateWithFlags(&e_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e_join, hipEventDisableTiming) != hipSuccess)
              return fail("event");
            // (a wait that silently failed would let the side stream's launches race the level's tables)
            if (hipEventRecord(e_fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)side, e_fork, 0) != hipSuccess)
              return fail("mprg_forest_level: fork onto the side stream");
          }
          auto loop = [&](int which, void *on) {
            return d_cluster_loop(FP(const int64_t, MPRG_F_SUB), FP(const int64_t, MPRG_F_PTAB), (int)cap_p, (int)F[MPRG_F_N_INIT], FP(const double, MPRG_F_UNIFORMS),
                                  (const int32_t *)(uintptr_t)F[MPRG_F_UOFF_HOST], FP(const double, MPRG_F_X), FP(double, MPRG_F_WS),
                                  FP(const int32_t, MPRG_F_D_OF_ROW), FP(const uint8_t, MPRG_F_GCODES), FP(int32_t, MPRG_F_CF_SCRATCH), FP(int32_t, MPRG_F_LABELS),
                                  FP(int32_t, MPRG_F_ASSIGN), FP(double, MPRG_F_KM_INFO), FP(int32_t, MPRG_F_KM_STATUS), FP(int32_t, MPRG_F_NUM_CLUSTERS),
                                  FP(int32_t, MPRG_F_ACTIVE), b4, which, on, dp);
          };
          if (split) {
            if (lds) {
              // the LDS classes on the side stream, beside them the general form for the problems no class holds; after both the general
              // form once more, for the (rare) problems that left the classes on the way
              if (loop(MPRG_LOOP_GENERAL | MPRG_LOOP_SKIP_LDS, stream) != 0) return -1;
              if (loop(MPRG_LOOP_LDS, side) != 0) return -1;
            } else {
              if (loop(forms & (MPRG_LOOP_GENERAL | MPRG_LOOP_SKIP_SMALL), stream) != 0) return -1;
              if (loop(forms & (MPRG_LOOP_SMALL_LOW | MPRG_LOOP_SMALL_HIGH), side) != 0) return -1;
            }
            if (hipEventRecord(e_join, (hipStream_t)side) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, e_join, 0) != hipSuccess)
              return fail("mprg_forest_level: join of the side stream");
            // (released by the runtime once the recorded work is done)
            if (hipEventDestroy(e_fork) != hipSuccess || hipEventDestroy(e_join) != hipSuccess) return fail("mprg_forest_level: event release");
            if (lds && loop(MPRG_LOOP_GENERAL, stream) != 0) return -1;
          } else if (loop(forms, stream) != 0) return -1;
        }
        // ---- S7 MultiClusterNodes and their children
        const long long cap_ns = F[MPRG_F_NSPLITS];
        const KfCheck ck5 = check(MPRG_STEP_SPLITS, 3, {cap_ns, BIG, C[MPRG_CAP_NCHILD]}, 1, MPRG_DS_POOL_USED, C[MPRG_CAP_POOL], 2, MPRG_DS_NNODES,
                                  C[MPRG_CAP_NODES]);
        const KfStep st5{ds, b5, &ck5};
        if (kf_splits_count(F, st5, b3 + 0, stream) != 0) return -1;
        if (cap_ns > 0) {
          if (kf_splits_fill(F, st5, b3 + 0, stream) != 0) return -1;
          LAUNCH(k_split_children, cap_ns, 64, stream, FP(const int64_t, MPRG_F_SUB), pool, FP(const int64_t, MPRG_F_SPT), FP(const int64_t, MPRG_F_SP),
                 FP(const int32_t, MPRG_F_D_OF_ROW), FP(const int32_t, MPRG_F_S_OF_ROW), FP(const int32_t, MPRG_F_ASSIGN), FP(int32_t, MPRG_F_POOL),
                 FP(int32_t, MPRG_F_CHILD_SIZES), slot(b5 + 0));
          if (kf_split_children(F, st5, b5 + 0, stream) != 0) return -1;
        }
      }
    }
  }
  LAUNCH(k_ds_advance, 1, 64, stream, ds, (const int64_t *)b1, (const int64_t *)b5);
  return check_launch("mprg_forest_level");
}

int mprg_forest_assemble_special(const int64_t *F, void *stream) {
  const long long n = F[MPRG_F_N_NODES];
  if (hipMemsetAsync(FHDR, 0, sizeof(int64_t) * MPRG_FOREST_HDR, (hipStream_t)stream) != hipSuccess) return fail("memset");
  if (n > 0) LAUNCH(k_as_special, KF_GRID(n), 256, stream, FP(const int64_t, MPRG_F_NODES), n, FP(const int32_t, MPRG_F_FAILED),
                    FP(int64_t, MPRG_F_SPECIAL_LIST), (int64_t)F[MPRG_F_SPECIAL_CAP], FHDR);
  return kf_publish(F, stream, "k_as_special");
}
int mprg_forest_assemble_layout(const int64_t *F, void *stream) {
  const long long n = F[MPRG_F_N_NODES], M = F[MPRG_F_N_MSAS];
  if (n <= 0) return 0;
  const int64_t *nodes = FP(const int64_t, MPRG_F_NODES), *root_of = FP(const int64_t, MPRG_F_ROOT_OF);
  const int32_t *failed = FP(const int32_t, MPRG_F_FAILED);
  int64_t *A = FP(int64_t, MPRG_F_ASM), *vm = FP(int64_t, MPRG_F_VALS_MSA), *vn = FP(int64_t, MPRG_F_VALS_NODE), *vp = FP(int64_t, MPRG_F_VALS_POS);
  int64_t *tmp = FP(int64_t, MPRG_F_SCAN_TMP), *mb = FP(int64_t, MPRG_F_MSA_BASE);
  const int64_t *LV = (const int64_t *)(uintptr_t)F[MPRG_F_LEVELS];
  const int nl = (int)F[MPRG_F_N_LEVELS];
  LAUNCH(k_as_init, KF_GRID(n), 256, stream, nodes, n, failed, A);
  if (F[MPRG_F_N_PATCH] > 0) LAUNCH(k_as_patch, KF_GRID(F[MPRG_F_N_PATCH]), 256, stream, (long long)F[MPRG_F_N_PATCH], FP(const int64_t, MPRG_F_PATCH), A);
  for (int l = nl - 2; l >= 0; --l) if (LV[4 * l + 1] > 0) LAUNCH(k_as_up, KF_GRID(LV[4 * l + 1]), 256, stream, nodes, (long long)LV[4 * l], (long long)LV[4 * l + 1], A, (int)A_SIZE);
  for (int l = 0; l + 1 < nl; ++l) if (LV[4 * l + 1] > 0) LAUNCH(k_as_pre, KF_GRID(LV[4 * l + 1]), 256, stream, nodes, (long long)LV[4 * l], (long long)LV[4 * l + 1], A);
  LAUNCH(k_as_tree_sizes, KF_GRID(M), 256, stream, M, root_of, (const int64_t *)A, vm);
  if (kf_scan(vm, M, 1, FHDR + 2, tmp, stream) != 0) return fail("scan");
  if (hipMemsetAsync(FP(int64_t, MPRG_F_N_SITES), 0, sizeof(int64_t) * M, (hipStream_t)stream) != hipSuccess) return fail("memset");
  LAUNCH(k_as_openers, KF_GRID(n), 256, stream, nodes, n, failed, (const int64_t *)A, (const int64_t *)vm, vp, FP(int64_t, MPRG_F_N_SITES));
  if (kf_scan(vp, n, 1, FHDR + 3, tmp, stream) != 0) return fail("scan");
  LAUNCH(k_as_sites, KF_GRID(n), 256, stream, nodes, n, failed, A, (const int64_t *)vm, (const int64_t *)vp);
  for (int l = nl - 2; l >= 0; --l) if (LV[4 * l + 1] > 0) LAUNCH(k_as_up, KF_GRID(LV[4 * l + 1]), 256, stream, nodes, (long long)LV[4 * l], (long long)LV[4 * l + 1], A, (int)A_TOTAL);
  LAUNCH(k_as_msa_len, KF_GRID(M), 256, stream, M, root_of, failed, (const int64_t *)A, mb);
  if (kf_scan(mb, M, 1, FHDR, tmp, stream) != 0) return fail("scan");
  LAUNCH(k_as_job_count, KF_GRID(n), 256, stream, nodes, n, (const int64_t *)A, (const int64_t *)vm, vn);
  if (kf_scan(vn, n, 1, FHDR + 1, tmp, stream) != 0) return fail("scan");
  return kf_publish(F, stream, "k_as_layout");
}
int mprg_forest_assemble_emit(const int64_t *F, void *stream) {
  const long long n = F[MPRG_F_N_NODES], M = F[MPRG_F_N_MSAS];
  if (n <= 0) return 0;
  const int64_t *nodes = FP(const int64_t, MPRG_F_NODES), *root_of = FP(const int64_t, MPRG_F_ROOT_OF);
  const int32_t *failed = FP(const int32_t, MPRG_F_FAILED);
  int64_t *A = FP(int64_t, MPRG_F_ASM);
  const int64_t *LV = (const int64_t *)(uintptr_t)F[MPRG_F_LEVELS];
  const int nl = (int)F[MPRG_F_N_LEVELS];
  LAUNCH(k_as_root_start, KF_GRID(M), 256, stream, M, root_of, failed, A, FP(int64_t, MPRG_F_MSA_BASE), FP(const int64_t, MPRG_F_VALS_MSA),
         FP(const int64_t, MPRG_F_VALS_NODE));
  for (int l = 0; l + 1 < nl; ++l) if (LV[4 * l + 1] > 0) LAUNCH(k_as_start, KF_GRID(LV[4 * l + 1]), 256, stream, nodes, (long long)LV[4 * l], (long long)LV[4 * l + 1], A);
  for (int l = 0; l < nl; ++l) if (LV[4 * l + 1] > 0)
    LAUNCH(k_as_leaf_jobs, KF_GRID(LV[4 * l + 1]), 256, stream, nodes, (long long)LV[4 * l], (long long)LV[4 * l + 1], failed, (const int64_t *)A,
           FP(const int64_t, MPRG_F_VALS_NODE), FP(const int64_t, MPRG_F_VALS_MSA), FP(const int64_t, MPRG_F_MSA_BASE),
           FP(const int64_t, MPRG_F_META), FP(const int32_t, MPRG_F_POOL),
           (const int32_t *)(uintptr_t)LV[4 * l + 2], (const int32_t *)(uintptr_t)LV[4 * l + 3], FP(int64_t, MPRG_F_JOBS), FP(uint8_t, MPRG_F_OUT),
           FP(int32_t, MPRG_F_INDEX_OUT));
  return check_launch("k_as_emit");
}
int mprg_forest_export_count(const int64_t *F, void *stream) {
  const long long n = F[MPRG_F_N_NODES];
  if (n <= 0) return 0;
  int64_t *vp = FP(int64_t, MPRG_F_VALS_POS);
  LAUNCH(k_ex_count, KF_GRID(n), 256, stream, FP(const int64_t, MPRG_F_NODES), n, FP(const int64_t, MPRG_F_ASM), FP(const int64_t, MPRG_F_VALS_MSA), vp);
  if (kf_scan(vp, n, 1, FHDR, FP(int64_t, MPRG_F_SCAN_TMP), stream) != 0) return fail("scan");
  return kf_publish(F, stream, "k_ex_count");
}
int mprg_forest_export_fill(const int64_t *F, void *stream) {
  const long long n = F[MPRG_F_N_NODES];
  if (n <= 0) return 0;
  LAUNCH(k_ex_fill, KF_GRID(n), 256, stream, FP(const int64_t, MPRG_F_NODES), n, FP(const int64_t, MPRG_F_ASM), FP(const int64_t, MPRG_F_VALS_MSA),
         FP(const int64_t, MPRG_F_VALS_POS), FP(const int32_t, MPRG_F_POOL), FP(int32_t, MPRG_F_EX_RECORDS), FP(int32_t, MPRG_F_EX_ROWS),
         FP(int64_t, MPRG_F_MSA_BASE), (long long)F[MPRG_F_N_MSAS]);
  return check_launch("k_ex_fill");
}
#undef FP
#undef FHDR
int mprg_export_alignments(const uint8_t *arena, const int64_t *meta, const int64_t *row_base, const int64_t *out_off, long long n_msas,
                           long long total_rows, uint8_t *out, void *stream) {
  if (n_msas <= 0 || total_rows <= 0) return 0;
  LAUNCH(k_ex_pack_rows, total_rows, 256, stream, arena, meta, row_base, out_off, n_msas, out);
  return check_launch("k_ex_pack_rows");
}

// numpy.random.RandomState(seed).random_sample(n): MT19937, init_genrand seeding, 53-bit doubles
void mprg_random_sample_host(uint32_t seed, int n, double *out) {
  uint32_t mt[624];
  mt[0] = seed;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int idx = 624;
  auto next = [&]() -> uint32_t {
    if (idx >= 624) {
      for (int i = 0; i < 624; i++) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  };
  for (int i = 0; i < n; i++) {
    const uint32_t a = next() >> 5, b = next() >> 6;
    out[i] = (a * 67108864.0 + b) / 9007199254740992.0;
  }
}

#ifdef KM_PHASE_TIMING
int mprg_debug_phase_cycles(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(km_phase_cycles), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(km_phase_cycles), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
}  // extern "C"
