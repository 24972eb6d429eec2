// mprg_platform.h — one kernel source, two builds.
//
//  * hipcc --offload-arch=gfx950 (the product): kernels are __global__ functions, PAR_FOR strides a workgroup's
//    threads over an index range, BARRIER is __syncthreads().
//  * g++ -DMPRG_CPU_EMU (tests only, tests/emu): the same kernel bodies run one workgroup at a time with PAR_FOR
//    as a plain loop, so the kernel LOGIC can be checked against the oracle in the GPU-less build container.
//    The emulation library is never loaded by the product (make_prg_amd loads libmprg_hip.so or fails).
//
// Kernel style this imposes: a kernel body is a sequence of PAR_FOR regions separated by BARRIER(); anything kept
// across a barrier lives in LDS (SHARED) or global memory; code outside a PAR_FOR is workgroup-uniform and any
// store there is guarded by ONE_THREAD.
#pragma once
#include <stdint.h>
#include <math.h>

#ifdef MPRG_CPU_EMU
#include <string.h>
#include <algorithm>
#define MPRG_DEV static inline
#define MPRG_DEVM inline
#define KERNEL(name, ...) static void name(int mprg_bid, int mprg_nthreads, __VA_ARGS__)
#define KERNEL_OCC(name, waves, ...) KERNEL(name, __VA_ARGS__)
#define LAUNCH(name, nblocks, nthreads, stream, ...)                          \
  do { for (int b_ = 0; b_ < (int)(nblocks); ++b_) name(b_, (int)(nthreads), __VA_ARGS__); } while (0)
#define BLOCK_ID mprg_bid
#define N_THREADS mprg_nthreads
#define PAR_FOR(i, n) for (long long i = 0; i < (long long)(n); ++i)
#define BARRIER() ((void)0)
#define ONE_THREAD if (true)
#define SHARED(T, name, n) T name[n]
typedef void *mprg_stream_t;
template <class T> MPRG_DEV T emu_atomic_or(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> MPRG_DEV T emu_atomic_max(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> MPRG_DEV T emu_atomic_min(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> MPRG_DEV T emu_atomic_add(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> MPRG_DEV T emu_atomic_cas(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
#define ATOMIC_OR(p, v) emu_atomic_or(p, v)
#define ATOMIC_MAX(p, v) emu_atomic_max(p, v)
#define ATOMIC_MIN(p, v) emu_atomic_min(p, v)
#define ATOMIC_ADD(p, v) emu_atomic_add(p, v)
#define ATOMIC_CAS(p, c, v) emu_atomic_cas(p, c, v)
#define FMA(a, b, c) fma(a, b, c)
#else
#include <hip/hip_runtime.h>
#define MPRG_DEV __device__ __forceinline__
#define MPRG_DEVM __device__ __forceinline__
#define KERNEL(name, ...) __global__ void name(__VA_ARGS__)
// at most 256 threads per workgroup (else the compiler must budget registers for 1024) and `waves` waves per SIMD
#define KERNEL_OCC(name, waves, ...) \
  __attribute__((amdgpu_flat_work_group_size(64, 256), amdgpu_waves_per_eu(waves, waves))) __global__ void name(__VA_ARGS__)
#define LAUNCH(name, nblocks, nthreads, stream, ...) \
  hipLaunchKernelGGL(name, dim3((unsigned)(nblocks)), dim3((unsigned)(nthreads)), 0, (hipStream_t)(stream), __VA_ARGS__)
#define BLOCK_ID ((int)blockIdx.x)
#define N_THREADS ((int)blockDim.x)
#define PAR_FOR(i, n) for (long long i = threadIdx.x; i < (long long)(n); i += blockDim.x)
#define BARRIER() __syncthreads()
#define ONE_THREAD if (threadIdx.x == 0)
#define SHARED(T, name, n) __shared__ T name[n]
typedef hipStream_t mprg_stream_t;
#define ATOMIC_OR(p, v) atomicOr(p, v)
#define ATOMIC_MAX(p, v) atomicMax(p, v)
#define ATOMIC_MIN(p, v) atomicMin(p, v)
#define ATOMIC_ADD(p, v) atomicAdd(p, v)
#define ATOMIC_CAS(p, c, v) atomicCAS(p, c, v)
#define FMA(a, b, c) __fma_rn(a, b, c)
#endif

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
#if defined(KM_PHASE_TIMING) && !defined(MPRG_CPU_EMU)
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

