// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets the GPU-less build container run the HIP kernel sources of
// make_prg_amd/csrc UNCHANGED on the CPU (g++ -I tests/emu/include), to check kernel logic and the host engine against
// the oracle.  The product is built by hipcc against the real header and never sees this file.
//
// Execution model: a launch runs its workgroups one after another; the threads of a workgroup are FIBERS (own stack,
// hand-rolled x86-64 context switch) of one OS thread.  A fiber runs until it reaches a synchronisation point:
//   __syncthreads()                      - completes when every live thread of the workgroup has arrived
//   __ballot / __any / __all / __shfl*   - complete when every live lane of the 64-wide wavefront is blocked; the lanes
//                                          blocked at a wave operation form its active set (they must all be at the same
//                                          operation with the same arguments — checked; the call SITE is not compared,
//                                          an optimising compiler duplicates call sites into both arms of a branch)
// A state in which nothing can complete (lanes of a wave split between different sync points for good) aborts with a
// message: wave-level operations must be reached by all live lanes of a wave that are not parked at a barrier or done.
// MPRG_EMU_ORDER=reverse runs the fibers of a workgroup in descending thread order (shakes out order dependence).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sys/mman.h>
#include <algorithm>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, name) type *name = (type *)emu::dyn_shared;
#define HIP_SYMBOL(x) x

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulation"; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
// memory / stream / event calls behind the mprg_rt_* plumbing: "device" memory is host memory, everything is synchronous
#include <chrono>
typedef void *hipEvent_t;
enum { hipErrorNotReady = 600, hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2 };
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : 101; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = calloc(1, sizeof(double)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  *(double *)e = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double *)b - *(double *)a); return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
// (diagnostic builds, -DKM_PHASE_TIMING: a "device symbol" is the host array itself)
template <class T> static inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }

namespace emu {
enum { RUNNABLE = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3 };
enum { OP_BALLOT = 1, OP_SHFL = 2, OP_SHFL_UP = 3, OP_SHFL_DOWN = 4, OP_SHFL_XOR = 5, OP_FIRST = 6 };
struct Fiber {
  void *sp;
  int state, op, arg, width;
  const void *site;
  uint64_t payload, result;
};
struct Uint3 { unsigned x, y, z; };
inline Uint3 tid, bid, bdim, gdim;
inline void *dyn_shared = nullptr;
inline Fiber *fibers = nullptr;
inline int cur = -1, n_fibers = 0;
inline void *sched_sp = nullptr;
inline char *stacks = nullptr;
inline const std::function<void()> *body = nullptr;
static const size_t STACK = 256 << 10;
static const int MAX_THREADS = 1024;

extern "C" void mprg_emu_switch(void **save_sp, void *load_sp);
#ifndef MPRG_EMU_SWITCH_DEFINED
#define MPRG_EMU_SWITCH_DEFINED
asm(R"(
.text
.globl mprg_emu_switch
.type mprg_emu_switch,@function
mprg_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size mprg_emu_switch,.-mprg_emu_switch
)");
#endif

inline void yield_to_scheduler() { mprg_emu_switch(&fibers[cur].sp, sched_sp); }
inline void fiber_main() {
  (*body)();
  fibers[cur].state = DONE;
  yield_to_scheduler();
  abort();
}
inline void die(const char *what) {
  fprintf(stderr, "hip emulation: %s (block %u, %d threads)\n", what, bid.x, n_fibers);
  for (int w = 0; w * 64 < n_fibers; ++w) {
    fprintf(stderr, "  wave %d:", w);
    for (int l = w * 64; l < n_fibers && l < w * 64 + 64; ++l) fprintf(stderr, " %d", fibers[l].state);
    fprintf(stderr, "\n");
  }
  abort();
}
inline void resolve_wave_group(int w0, int w1, const void *) {
  // the lanes of [w0, w1) blocked at a wave operation are its active set; they must agree on the operation
  uint64_t active = 0;
  for (int l = w0; l < w1; ++l) if (fibers[l].state == AT_WAVE) active |= 1ull << (l - w0);
  int first = __builtin_ctzll(active);
  for (int l = w0; l < w1; ++l)
    if (((active >> (l - w0)) & 1) && (fibers[l].op != fibers[w0 + first].op || fibers[l].width != fibers[w0 + first].width ||
                                       (fibers[l].op != OP_SHFL && fibers[l].arg != fibers[w0 + first].arg)))
      die("divergent wave operation: the lanes of a wave are blocked at different ballots / shuffles");
  uint64_t ballot = 0;
  for (int l = w0; l < w1; ++l) if ((active >> (l - w0)) & 1) if (fibers[l].payload) ballot |= 1ull << (l - w0);
  for (int l = w0; l < w1; ++l) {
    if (!((active >> (l - w0)) & 1)) continue;
    Fiber &f = fibers[l];
    const int lane = l - w0, width = f.width > 0 ? f.width : 64, seg = lane & ~(width - 1);
    int src = lane;
    switch (f.op) {
      case OP_BALLOT: f.result = ballot; break;
      case OP_FIRST: f.result = fibers[w0 + first].payload; break;
      case OP_SHFL: src = seg + (f.arg & (width - 1)); break;
      case OP_SHFL_UP: src = lane - f.arg >= seg ? lane - f.arg : lane; break;
      case OP_SHFL_DOWN: src = lane + f.arg < seg + width ? lane + f.arg : lane; break;
      case OP_SHFL_XOR: src = (lane ^ f.arg); if (src >= seg + width) src = lane; break;
    }
    if (f.op >= OP_SHFL && f.op <= OP_SHFL_XOR) {
      // reading an inactive lane is undefined on hardware; the emulation returns the lane's own value
      const bool ok = w0 + src < w1 && ((active >> src) & 1);
      f.result = ok ? fibers[w0 + src].payload : f.payload;
    }
  }
  for (int l = w0; l < w1; ++l) if ((active >> (l - w0)) & 1) fibers[l].state = RUNNABLE;
}

inline void run_block(const std::function<void()> &fn, int nthreads) {
  if (nthreads > MAX_THREADS) die("more than 1024 threads per workgroup");
  if (!stacks) {
    stacks = (char *)mmap(nullptr, STACK * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    fibers = new Fiber[MAX_THREADS];
  }
  static const bool reverse = getenv("MPRG_EMU_ORDER") && !strcmp(getenv("MPRG_EMU_ORDER"), "reverse");
  body = &fn;
  n_fibers = nthreads;
  for (int t = 0; t < nthreads; ++t) {
    uint64_t *top = (uint64_t *)(stacks + STACK * (t + 1));
    top[-1] = 0;
    top[-2] = (uint64_t)(void *)&fiber_main;
    for (int q = 3; q <= 8; ++q) top[-q] = 0;
    fibers[t].sp = top - 8;
    fibers[t].state = RUNNABLE;
  }
  for (;;) {
    bool ran = false;
    int n_done = 0, n_bar = 0;
    for (int i = 0; i < nthreads; ++i) {
      const int t = reverse ? nthreads - 1 - i : i;
      if (fibers[t].state == RUNNABLE) {
        cur = t;
        tid.x = (unsigned)t % bdim.x; tid.y = ((unsigned)t / bdim.x) % bdim.y; tid.z = (unsigned)t / (bdim.x * bdim.y);
        mprg_emu_switch(&sched_sp, fibers[t].sp);
        ran = true;
      }
      n_done += fibers[t].state == DONE;
      n_bar += fibers[t].state == AT_BARRIER;
    }
    if (n_done == nthreads) break;
    // wave operations whose wave has no runnable lane left
    bool released = false;
    for (int w0 = 0; w0 < nthreads; w0 += 64) {
      const int w1 = std::min(w0 + 64, nthreads);
      bool blocked = true, any_wave = false;
      const void *site = nullptr;
      for (int l = w0; l < w1; ++l) {
        if (fibers[l].state == RUNNABLE) blocked = false;
        if (fibers[l].state == AT_WAVE && !any_wave) { any_wave = true; site = fibers[l].site; }
      }
      if (blocked && any_wave) { resolve_wave_group(w0, w1, site); released = true; }
    }
    if (released) continue;
    if (n_bar > 0 && n_bar + n_done == nthreads) {
      for (int t = 0; t < nthreads; ++t) if (fibers[t].state == AT_BARRIER) fibers[t].state = RUNNABLE;
      continue;
    }
    if (!ran) die("deadlock: no thread can make progress");
  }
  cur = -1;
}

__attribute__((noinline)) inline uint64_t wave_op(int op, uint64_t payload, int arg, int width, const void *site) {
  Fiber &f = fibers[cur];
  f.op = op; f.payload = payload; f.arg = arg; f.width = width; f.site = site; f.state = AT_WAVE;
  yield_to_scheduler();
  return fibers[cur].result;
}
template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
}  // namespace emu

#define threadIdx emu::tid
#define blockIdx emu::bid
#define blockDim emu::bdim
#define gridDim emu::gdim
#define warpSize 64

__attribute__((noinline)) static void __syncthreads() {
  emu::fibers[emu::cur].state = emu::AT_BARRIER;
  emu::yield_to_scheduler();
}
#define MPRG_EMU_SITE() __builtin_extract_return_addr(__builtin_return_address(0))
__attribute__((noinline)) static unsigned long long __ballot(int pred) { return emu::wave_op(emu::OP_BALLOT, pred != 0, 0, 64, MPRG_EMU_SITE()); }
__attribute__((noinline)) static int __any(int pred) { return emu::wave_op(emu::OP_BALLOT, pred != 0, 0, 64, MPRG_EMU_SITE()) != 0; }
__attribute__((noinline)) static int __all(int pred) { return emu::wave_op(emu::OP_BALLOT, pred == 0, 0, 64, MPRG_EMU_SITE()) == 0; }
static inline unsigned long long __activemask() { return __ballot(1); }
#define MPRG_EMU_SHFL(T)                                                                                                    \
  __attribute__((noinline)) static T __shfl(T v, int src, int width = 64) { return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL, emu::to_bits(v), src, width, MPRG_EMU_SITE())); } \
  __attribute__((noinline)) static T __shfl_up(T v, unsigned d, int width = 64) { return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL_UP, emu::to_bits(v), (int)d, width, MPRG_EMU_SITE())); } \
  __attribute__((noinline)) static T __shfl_down(T v, unsigned d, int width = 64) { return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL_DOWN, emu::to_bits(v), (int)d, width, MPRG_EMU_SITE())); } \
  __attribute__((noinline)) static T __shfl_xor(T v, int m, int width = 64) { return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL_XOR, emu::to_bits(v), m, width, MPRG_EMU_SITE())); }
MPRG_EMU_SHFL(int)
MPRG_EMU_SHFL(unsigned)
MPRG_EMU_SHFL(long long)
MPRG_EMU_SHFL(unsigned long long)
MPRG_EMU_SHFL(float)
MPRG_EMU_SHFL(double)
// DPP move, quad_perm controls (0x00 .. 0xff): lane L reads lane (ctrl >> 2 (L & 3)) & 3 of its quad
static inline int __builtin_amdgcn_mov_dpp(int v, int ctrl, int, int, bool) { return __shfl(v, (ctrl >> (2 * ((int)threadIdx.x & 3))) & 3, 4); }
__attribute__((noinline)) static int __builtin_amdgcn_readfirstlane(int v) { return (int)emu::wave_op(emu::OP_FIRST, (uint64_t)(uint32_t)v, 0, 64, MPRG_EMU_SITE()); }
static inline int __lane_id() { return (int)(emu::cur & 63); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline double __fma_rn(double a, double b, double c) { return fma(a, b, c); }
static inline long long clock64() { return 0; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// atomics: one OS thread runs every fiber, so plain read-modify-write is atomic
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

template <class K, class... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t dyn_bytes, hipStream_t, A... args) {
  std::vector<uint64_t> dyn((dyn_bytes + 7) / 8 + 1);
  emu::dyn_shared = dyn.data();
  emu::bdim = {block.x, block.y, block.z};
  emu::gdim = {grid.x, grid.y, grid.z};
  const std::function<void()> fn = [&]() { kernel(args...); };
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        emu::bid = {bx, by, bz};
        memset(dyn.data(), 0xA5, dyn.size() * 8);          // poison: LDS is not initialised on hardware either
        emu::run_block(fn, (int)(block.x * block.y * block.z));
      }
  emu::dyn_shared = nullptr;
}
