// mprg_platform.h — shared helpers of the gfx950 kernels (HIP only; built by hipcc --offload-arch=gfx950).
//
// Loop / launch shorthands used throughout csrc/: PAR_FOR strides the threads of a workgroup over an index range,
// BARRIER is __syncthreads(), ONE_THREAD guards workgroup-uniform stores.  Wave-level code (ballot, shuffles, DPP-style
// scans) is written with the HIP intrinsics directly — a wavefront is 64 lanes on CDNA4; the WAVE_* helpers below are
// the 64-lane idioms several kernels share.
// (The GPU-less build container checks kernel logic by compiling these same sources with g++ against a test-only
// stand-in for <hip/hip_runtime.h> that runs workgroups as fibers, tests/emu/include; nothing here knows about it.)
#pragma once
#include <stdint.h>
#include <math.h>
#include <hip/hip_runtime.h>

#define MPRG_DEV __device__ __forceinline__
#define MPRG_DEVM __device__ __forceinline__
#define KERNEL(name, ...) __global__ void name(__VA_ARGS__)
// `waves` waves per SIMD (fixes the register budget: 512 / waves VGPRs), workgroups of up to 64 * 4 * waves threads
#define KERNEL_OCC(name, waves, ...) \
  __attribute__((amdgpu_flat_work_group_size(64, 256 * (waves)), amdgpu_waves_per_eu(waves, waves))) __global__ void name(__VA_ARGS__)
#define LAUNCH(name, nblocks, nthreads, stream, ...) \
  hipLaunchKernelGGL(name, dim3((unsigned)(nblocks)), dim3((unsigned)(nthreads)), 0, (hipStream_t)(stream), __VA_ARGS__)
#define LAUNCH_LDS(name, nblocks, nthreads, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(name, dim3((unsigned)(nblocks)), dim3((unsigned)(nthreads)), (size_t)(lds_bytes), (hipStream_t)(stream), __VA_ARGS__)
// a (nx, ny) grid: blockIdx.y = which PART of item blockIdx.x's work the workgroup takes (kernels whose items can be huge)
#define LAUNCH2(name, nx, ny, nthreads, stream, ...) \
  hipLaunchKernelGGL(name, dim3((unsigned)(nx), (unsigned)(ny)), dim3((unsigned)(nthreads)), 0, (hipStream_t)(stream), __VA_ARGS__)
#define BLOCK_ID ((int)blockIdx.x)
#define N_THREADS ((int)blockDim.x)
// MPRG_TID: the thread's index as PAR_FOR / ONE_THREAD see it — a FRESH copy at every use (one v_mov the compiler cannot see
// through).  Why: a kernel that loops over a big inlined body (k_kloop.inc: the rounds of a clustering problem around ~30 000
// instructions) would otherwise compute everything the body derives from the thread index once, before the loop, and keep it —
// spilled to scratch and reloaded — across the whole body (241 scratch reloads per fit against 53; the reloads go through the
// vector L1's address unit, the block the KMeans kernels saturate).
#if defined(__HIP_DEVICE_COMPILE__)
MPRG_DEV int mprg_tid_fresh() { int t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"((int)threadIdx.x)); return t; }
#define MPRG_TID mprg_tid_fresh()
#else
#define MPRG_TID ((int)threadIdx.x)
#endif
#define PAR_FOR(i, n) for (long long i = MPRG_TID; i < (long long)(n); i += blockDim.x)
#define BARRIER() __syncthreads()
#define ONE_THREAD if (MPRG_TID == 0)
#define SHARED(T, name, n) __shared__ T name[n]
typedef hipStream_t mprg_stream_t;
#define ATOMIC_OR(p, v) atomicOr(p, v)
#define ATOMIC_MAX(p, v) atomicMax(p, v)
#define ATOMIC_MIN(p, v) atomicMin(p, v)
#define ATOMIC_ADD(p, v) atomicAdd(p, v)
#define ATOMIC_CAS(p, c, v) atomicCAS(p, c, v)
#define FMA(a, b, c) __fma_rn(a, b, c)

// the value lane `src` of the wavefront holds (src wave-uniform), in every lane: two v_readlane — no trip through the LDS crossbar
// (__shfl is ds_bpermute), which matters where a wavefront folds 64 values one after the other
#if defined(__HIP_DEVICE_COMPILE__)
MPRG_DEV double wave_bcast(double v, int src) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#else
MPRG_DEV double wave_bcast(double v, int src) { return __shfl(v, src); }
#endif

// ---- 64-lane wavefront idioms
#define WAVE 64
MPRG_DEV int wave_lane() { return (int)(threadIdx.x & (WAVE - 1)); }
MPRG_DEV int wave_id() { return (int)(threadIdx.x >> 6); }
// number of lanes below this one whose predicate holds, and (out) the wave's total: one ballot, two popcounts
MPRG_DEV int wave_rank(bool pred, int *total) {
  const unsigned long long b = __ballot(pred);
  *total = __popcll(b);
  return __popcll(b & ((1ull << wave_lane()) - 1ull));
}
// inclusive prefix sum over the lanes of a wave (all 64 lanes must call it)
MPRG_DEV int wave_scan_incl(int v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) { const int y = __shfl_up(v, d); if (wave_lane() >= d) v += y; }
  return v;
}
MPRG_DEV long long wave_scan_incl_ll(long long v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) { const long long y = __shfl_up(v, d); if (wave_lane() >= d) v += y; }
  return v;
}
// Lanes of ONE wavefront that exchange data through LDS without a workgroup barrier (several small work items per workgroup, a
// wavefront each): the wavefront runs in lockstep, what has to be kept is the ORDER of its LDS accesses — the compiler must not
// move them across this point and the earlier ones must have completed: a wavefront-scope fence.  Every lane of the wavefront
// must reach it (converged control flow).
#if defined(__HIP_DEVICE_COMPILE__)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// ... and through GLOBAL memory (a wavefront's stores before its other lanes' loads of them): workgroup scope — the CU's vector
// L1 is shared by the workgroup's wavefronts, so this is a wait for the stores, not a cache operation
#define WAVE_SYNC_GLOBAL() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#else
#define WAVE_SYNC() ((void)__ballot(1))
#define WAVE_SYNC_GLOBAL() ((void)__ballot(1))
#endif
// exclusive prefix sum over ALL threads of a workgroup of up to 1024 threads (every thread calls it, uniform control
// flow); *total receives the workgroup's sum.  scratch: 17 ints of LDS, reusable after the call returns.
MPRG_DEV int block_scan_excl(int v, int *scratch, int *total) {
  const int incl = wave_scan_incl(v);
  const int nw = (int)((blockDim.x + WAVE - 1) >> 6);
  if (wave_lane() == WAVE - 1 || threadIdx.x == blockDim.x - 1) scratch[wave_id()] = incl;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int w = 0; w < nw; ++w) { const int t = scratch[w]; scratch[w] = run; run += t; } scratch[16] = run; }
  __syncthreads();
  const int out = scratch[wave_id()] + incl - v;
  *total = scratch[16];
  __syncthreads();
  return out;
}

// DPP quad permute of a double: lane L of every quad (lanes 4m .. 4m+3) reads the quad's lane (CTRL >> 2 (L & 3)) & 3; the
// lanes it reads must be active.  One VALU move per 32-bit half, no LDS crossbar.
template <int CTRL> MPRG_DEV double quad_perm(double v) {
  int w[2];
  __builtin_memcpy(w, &v, 8);
  w[0] = __builtin_amdgcn_mov_dpp(w[0], CTRL, 0xf, 0xf, true);
  w[1] = __builtin_amdgcn_mov_dpp(w[1], CTRL, 0xf, 0xf, true);
  double r;
  __builtin_memcpy(&r, w, 8);
  return r;
}
// 8 consecutive doubles at p fetched by a GROUP of G lanes (G = 4: a quad, G = 2: a pair — lanes 2m, 2m+1) that all run the
// same item: lane g of the group loads its 8 / G doubles, the pieces go round by quad_perm, every lane ends with all 8.
// The address unit of the vector L1 walks one cache line per lane and load instruction whatever the lane takes from it
// (24 lines per load instruction measured, profiles/r03/kmeans_counters.md): 1 / G of the load instructions per row is 1 / G of
// its line-cycles.
template <int G> MPRG_DEV void group_fetch8(const double *p, int g, double *out);
template <> MPRG_DEV void group_fetch8<4>(const double *p, int g, double *out) {
  const double l0 = p[2 * g], l1 = p[2 * g + 1];
  out[0] = quad_perm<0x00>(l0); out[1] = quad_perm<0x00>(l1); out[2] = quad_perm<0x55>(l0); out[3] = quad_perm<0x55>(l1);
  out[4] = quad_perm<0xaa>(l0); out[5] = quad_perm<0xaa>(l1); out[6] = quad_perm<0xff>(l0); out[7] = quad_perm<0xff>(l1);
}
template <> MPRG_DEV void group_fetch8<2>(const double *p, int g, double *out) {
  const double l0 = p[4 * g], l1 = p[4 * g + 1], l2 = p[4 * g + 2], l3 = p[4 * g + 3];
  // quad lanes {0, 1} read lane h, lanes {2, 3} read lane 2 + h: [h, h, 2 + h, 2 + h]
  out[0] = quad_perm<0xa0>(l0); out[1] = quad_perm<0xa0>(l1); out[2] = quad_perm<0xa0>(l2); out[3] = quad_perm<0xa0>(l3);
  out[4] = quad_perm<0xf5>(l0); out[5] = quad_perm<0xf5>(l1); out[6] = quad_perm<0xf5>(l2); out[7] = quad_perm<0xf5>(l3);
}

// ---- device-side counts (mprg_forest_level: a recursion level enqueued without a host wait)
// The host sizes such a launch from a CAPACITY; the exact count is a word of the forest's device state `ds` that an earlier kernel
// of the same stream wrote.  Workgroups / items beyond it return at once, and so does every kernel once an earlier step's totals
// exceeded their capacity (ds[0], sticky: the host then repeats the forest with exact sizes).  ds == nullptr: the host's count is exact.
struct DsCount { const int64_t *ds; int slot; };
#define DS_HOST DsCount{nullptr, 0}
MPRG_DEV long long ds_n(const DsCount c, long long host_n) { return c.ds ? (c.ds[0] ? 0 : (long long)c.ds[c.slot]) : host_n; }
// first statement of a kernel whose workgroup b handles items [b * per, (b + 1) * per)
#define DS_GUARD(dc, per) do { if ((dc).ds && (long long)BLOCK_ID * (per) >= ds_n(dc, 0)) return; } while (0)

// ---- cell codes and view accessors (layout: include/mprg.h)
#define C_GAP 4
#define C_N 11
#define BIT_GAP (1u << 4)
#define BIT_N (1u << 11)
#define BITS_IUPAC 0x7E0u
#define VF 12  // MPRG_VIEW_FIELDS
#define PF 12  // MPRG_PROB_FIELDS

struct ViewD {
  long long rm, cm;
  int pitchC, pitchS, rows_off, n_rows, col0, n_cols;
  long long col_off, row_off, aux0, aux1;
};
MPRG_DEV ViewD load_view(const int64_t *views, int v) {
  const int64_t *p = views + (long long)v * VF;
  ViewD d;
  d.rm = p[0]; d.cm = p[1]; d.pitchC = (int)p[2]; d.pitchS = (int)p[3]; d.rows_off = (int)p[4];
  d.n_rows = (int)p[5]; d.col0 = (int)p[6]; d.n_cols = (int)p[7]; d.col_off = p[8]; d.row_off = p[9];
  d.aux0 = p[10]; d.aux1 = p[11];
  return d;
}
MPRG_DEV int view_row(const ViewD &d, const int32_t *rowidx, long long i) {
  return d.rows_off < 0 ? (int)i : rowidx[d.rows_off + i];
}
// cell of MSA row `row`, view column c, read from the transposed copy (coalesced when threads differ in row)
MPRG_DEV unsigned cell_t(const uint8_t *arena, const ViewD &d, int row, int c) {
  return arena[d.cm + (long long)(d.col0 + c) * d.pitchS + row];
}
// consensus decision of one column from its presence mask: symbol code 0..3 or 255 for '*'
// (utils/seq_utils.py:228-238: N ignored; ambiguous bases, several symbols or only gaps -> '*')
MPRG_DEV int consensus_code(uint32_t mask) {
  uint32_t m = mask & ~BIT_N;
  if (m == 0 || (m & (m - 1)) != 0 || (m & BITS_IUPAC) || m == BIT_GAP) return 255;
  int c = 0;
  while (!((m >> c) & 1u)) ++c;
  return c;
}

// ---- diagnostic build only (-DKM_PHASE_TIMING, tools/phase_timing.py): shader-clock cycles thread 0 of every workgroup
// spends between marks, summed per mark id (0-15: k_kmeans_restart, 16-31: k_partition)
#if defined(KM_PHASE_TIMING)
__device__ unsigned long long km_phase_cycles[32];
#define KM_T0() long long km_t0 = clock64()
#define KM_TPARAM , long long &km_t0
#define KM_TARG , km_t0
#define KM_T(id) do { if (threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&km_phase_cycles[id], (unsigned long long)(now_ - km_t0)); km_t0 = now_; } } while (0)
#else
#define KM_T0() ((void)0)
#define KM_TPARAM
#define KM_TARG
#define KM_T(id) ((void)0)
#endif

