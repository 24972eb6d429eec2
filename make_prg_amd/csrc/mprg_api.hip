// mprg_api.hip — C ABI (include/mprg.h) over the gfx950 kernels of the from_msa hot path.
// Build (product):  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared mprg_api.hip -o libmprg_hip.so
#include "mprg_platform.h"
#include "../../include/mprg.h"
#include <stdio.h>
#include <string.h>

#include "k_ingest.inc"
#include "k_columns.inc"
#include "k_partition.inc"
#include "k_rows.inc"
#include "k_kmer.inc"
#include "k_kmeans.inc"
#include "k_cluster.inc"
#include "k_emit.inc"
#include "host_encoders.inc"

static thread_local char g_err[512] = "";
static int fail(const char *what) { snprintf(g_err, sizeof g_err, "%s", what); return -1; }

static int check_launch(const char *name) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(g_err, sizeof g_err, "%s: %s", name, hipGetErrorString(e)); return -2; }
  return 0;
}

#define BLOCK_VIEW 256
#include <stdlib.h>
static int env_threads(const char *name, int dflt) {
  const char *e = getenv(name);
  const int v = e ? atoi(e) : dflt;
  return (v >= 64 && v <= 1024 && v % 64 == 0) ? v : dflt;
}

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

int mprg_column_masks(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int32_t *work,
                      int n_items, int rows_per_chunk, uint32_t *out_mask, void *stream) {
  if (n_items <= 0) return 0;
  if (rows_per_chunk <= 0) return fail("rows_per_chunk must be positive");
  LAUNCH(k_column_masks, n_items, CM_THREADS, stream, arena, views, rowidx, work, rows_per_chunk, out_mask);
  return check_launch("k_column_masks");
}

int mprg_partition(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views,
                   const uint32_t *mask, int min_match_length, const int32_t *work_rows, int n_work_rows,
                   uint32_t *maxrun, int32_t *stack, int32_t *ivflag, int32_t *iv, int32_t *n_iv, int32_t *status,
                   int32_t *view_out, int32_t *iv_packed, int32_t *iv_count, const int32_t *fused_list, int n_fused,
                   const int32_t *other_list, int n_other, void *stream) {
  if (n_views <= 0) return 0;
  if (!fused_list && !other_list) { n_fused = 0; n_other = n_views; }
  else if (n_fused + n_other != n_views) return fail("mprg_partition: the two view lists must cover the views");
  if (n_work_rows > 0) LAUNCH(k_gap_runs, n_work_rows, GR_ROWS, stream, arena, views, rowidx, work_rows, mask, maxrun);
  if (n_other > 0)
    LAUNCH(k_partition, n_other, BLOCK_VIEW, stream, other_list, arena, views, rowidx, mask, min_match_length, maxrun, stack,
           ivflag, iv, n_iv, status, view_out);
  if (n_fused > 0)
    LAUNCH(k_partition_fused, n_fused, env_threads("MPRG_PF_THREADS", BLOCK_VIEW), stream, fused_list, arena, views, rowidx, min_match_length, iv, n_iv, status,
           view_out);
  if (view_out) {                                  // the packed list of all triples of the call
    if (!iv_packed || !iv_count) return fail("mprg_partition: view_out needs iv_packed and iv_count");
    LAUNCH(k_pack_scan, 1, 1024, stream, n_views, n_iv, view_out, iv_count);
    LAUNCH(k_pack_copy, (n_views + PK_THREADS / WAVE - 1) / (PK_THREADS / WAVE), PK_THREADS, stream, n_views, views, iv, view_out,
           iv_packed);
  }
  return check_launch("k_partition");
}

int mprg_ungap_dedupe(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, int n_views, int kmer_size,
                      const int32_t *work_rows, int n_work_rows, uint8_t *ucodes, uint64_t *hashes, int32_t *ulen,
                      int32_t *rep_u, int32_t *rep_g, int32_t *d_of_row, int32_t *s_of_row, int32_t *reps_pos,
                      int32_t *reps_len, int32_t *seqrow, int64_t *occ_off, int64_t *summary, uint8_t *gcodes, void *stream) {
  if (n_views <= 0) return 0;
  LAUNCH(k_ungap_hash, n_work_rows, UG_ROWS, stream, arena, views, rowidx, work_rows, ucodes, hashes, ulen, gcodes);
  LAUNCH(k_ungap_dedupe, n_views, env_threads("MPRG_DD_THREADS", BLOCK_VIEW), stream, arena, views, rowidx, kmer_size, ucodes, (const uint8_t *)gcodes, hashes, ulen, rep_u, rep_g,
         d_of_row, s_of_row, reps_pos, reps_len, seqrow, occ_off, summary);
  return check_launch("k_ungap_dedupe");
}

int mprg_kmer_dictionary(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size,
                         const uint8_t *ucodes, const int32_t *ulen, const int32_t *seqrow, int64_t *occ_off,
                         uint8_t *table, uint8_t *first_flag, int32_t *out_V, void *stream) {
  (void)ulen;
  if (n_probs <= 0) return 0;
  if (kmer_size < 1 || kmer_size > 16) return fail("k-mer size must be in 1..16 (4-bit packed keys)");
  LAUNCH(k_kmer_dictionary, n_probs, KD_THREADS, stream, views, prob, kmer_size, ucodes, seqrow, (const int64_t *)occ_off, table,
         first_flag, out_V);
  return check_launch("k_kmer_dictionary");
}

int mprg_kmer_counts(const int64_t *views, const int64_t *prob, int n_probs, int kmer_size, const uint8_t *ucodes,
                     const int32_t *ulen, const int32_t *seqrow, const int64_t *occ_off, const uint8_t *table,
                     double *xcounts, void *stream) {
  (void)ulen;
  if (n_probs <= 0) return 0;
  if (kmer_size < 1 || kmer_size > 16) return fail("k-mer size must be in 1..16 (4-bit packed keys)");
  LAUNCH(k_kmer_counts, n_probs, 512, stream, views, prob, kmer_size, ucodes, seqrow, occ_off, table, xcounts);
  return check_launch("k_kmer_counts");
}

int64_t mprg_kmeans_workspace_doubles(int64_t D, int64_t V, int k_max, int n_restart_slots) {
  if (k_max > KM_KMAX || V > (128LL << (KM_PW_DEPTH - 1))) return -1;       // V: depth of the pairwise-sum stack (k_kmeans.inc)
  return km_common_doubles_host(D, V) + (int64_t)n_restart_slots * km_restart_doubles_host(D, V);
}

int mprg_kmeans_prepare(const int64_t *prob, int n_probs, const double *xcounts, double *ws, const int32_t *lds_list,
                        int n_lds, int64_t lds_bytes, const int32_t *other_list, int n_other, void *stream) {
  if (n_probs <= 0) return 0;
  if (!lds_list && !other_list) { n_lds = 0; n_other = n_probs; }
  else if (n_lds + n_other != n_probs) return fail("mprg_kmeans_prepare: the two problem lists must cover the problems");
  if (n_lds > 0 && (lds_bytes <= 0 || lds_bytes > MPRG_KMEANS_PREPARE_LDS_MAX)) return fail("mprg_kmeans_prepare: lds_bytes out of range");
  if (n_other > 0) {
    LAUNCH(k_kmeans_prepare, n_other, 256, stream, other_list, prob, xcounts, ws);
    LAUNCH(k_kmeans_prepare_tables, (long long)n_other * KP_PARTS, 256, stream, other_list, prob, xcounts, ws);
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
    const int prep_threads = env_threads("MPRG_KP_THREADS", lds_bytes > 64 * 1024 ? 1024 : (lds_bytes > 24 * 1024 ? 512 : 256));
    LAUNCH_LDS(k_kmeans_prepare_lds, n_lds, prep_threads, lds_bytes, stream, lds_list, prob, xcounts, ws);
  }
  return check_launch("k_kmeans_prepare");
}

int mprg_kmeans_restarts(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *uniforms_dev,
                         const double *xcounts, double *ws, int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init > KM_RMAX) return fail("n_init must be <= 16");
  LAUNCH(k_kmeans_restart, n_fits, env_threads("MPRG_KM_THREADS", 256), stream, prob, kinfo, n_init, uniforms_dev, xcounts,
         ws, km_status);
  return check_launch("k_kmeans_restart");
}

int mprg_kmeans_fit(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *uniforms_dev,
                    const double *xcounts, double *ws, double *slot_ws, int64_t slot_stride_doubles, int n_slots,
                    int32_t *next_fit, int32_t *labels, double *km_info, int32_t *km_status, void *stream) {
  if (n_fits <= 0) return 0;
  if (n_init > KM_RMAX) return fail("n_init must be <= 16");
  static_assert(KM_SEL_LABELS * sizeof(int32_t) <= KM_XL_BYTES, "the selection's label staging lives in the restarts' LDS pool");
  if (!slot_ws) {                      // no scratch slots: one workgroup per fit on the restart regions of the problems' workspaces
    LAUNCH(k_kmeans_restart_select, n_fits, env_threads("MPRG_KM_THREADS", 256), stream, prob, kinfo, n_init, uniforms_dev, xcounts,
           ws, labels, km_info, km_status);
    return check_launch("k_kmeans_restart_select");
  }
  if (n_slots <= 0 || slot_stride_doubles <= 0) return fail("mprg_kmeans_fit: no scratch slots");
  if (hipMemsetAsync(next_fit, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return fail("memset");
  const int grid = n_fits < n_slots ? n_fits : n_slots;
  LAUNCH(k_kmeans_fit, grid, env_threads("MPRG_KM_THREADS", 256), stream, prob, kinfo, n_fits, n_init, uniforms_dev, xcounts, ws,
         slot_ws, (long long)slot_stride_doubles, next_fit, labels, km_info, km_status);
  return check_launch("k_kmeans_fit");
}

int mprg_kmeans_select(const int64_t *prob, const int32_t *kinfo, int n_fits, int n_init, const double *xcounts,
                       double *ws, int32_t *labels, double *km_info, void *stream) {
  if (n_fits <= 0) return 0;
  LAUNCH(k_kmeans_select, n_fits, 256, stream, prob, kinfo, n_init, xcounts, ws, labels, km_info);
  return check_launch("k_kmeans_select");
}

int mprg_cluster_further(const uint8_t *arena, const int64_t *views, const int32_t *rowidx, const int64_t *prob,
                         int n_probs, int k, const int32_t *d_of_row, const int32_t *labels, int32_t *assign,
                         const int32_t *work_cols, int n_work_cols, const int32_t *work_rows, int n_work_rows,
                         int32_t *scratch, int32_t *out_further, const double *km_info, const uint8_t *gcodes, void *stream) {
  if (n_probs <= 0) return 0;
  if (k < 1 || k > KM_KMAX) return fail("mprg_cluster_further: k out of range");
  if (hipMemsetAsync(out_further, 0, sizeof(int32_t) * n_probs, (hipStream_t)stream) != hipSuccess) return fail("memset");
  LAUNCH(k_cluster_majority, n_work_cols, CF_THREADS, stream, arena, views, rowidx, prob, work_cols, k, d_of_row, labels,
         assign, km_info, scratch, gcodes);
  LAUNCH(k_cluster_hamming, n_work_rows, CF_TILE, stream, arena, views, rowidx, prob, work_rows, d_of_row, labels,
         (const int32_t *)scratch, out_further, gcodes);
  return check_launch("k_cluster_further");
}

int mprg_split_children(const int64_t *views, const int32_t *rowidx, const int64_t *prob, int n_probs,
                        const int64_t *split_info, const int32_t *d_of_row, const int32_t *s_of_row,
                        const int32_t *assign, int32_t *pool_out, int32_t *child_sizes, void *stream) {
  if (n_probs <= 0) return 0;
  LAUNCH(k_split_children, n_probs, 64, stream, views, rowidx, prob, split_info, d_of_row, s_of_row, assign, pool_out,
         child_sizes);
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
