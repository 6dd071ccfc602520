/*
 * tests/emu/include/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A tiny host emulation of the subset of the HIP runtime + device language that
 * respmon_amd/csrc uses, so that the product's kernels and host orchestration can be
 * compiled with g++ and exercised for index/border/ordering logic on a machine without a
 * GPU (tests/test_emu_*.py).  It shadows <hip/hip_runtime.h> by include path in
 * tests/emu/build.py ONLY; the product library is always built by hipcc for gfx950 and the
 * product package never loads the emulated build.
 *
 * Model: blocks run one after another on the calling thread; the threads of a block are
 * ucontext fibers resumed round-robin, __syncthreads() is a yield; __shared__ variables are
 * function-local statics (shared by the block's threads because blocks are sequential).
 * "Device memory" is host memory.
 */
#ifndef RM_HIPEMU_RUNTIME_H
#define RM_HIPEMU_RUNTIME_H

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <ucontext.h>
#include <vector>

#define RM_HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

namespace hipemu {
struct Fiber { ucontext_t ctx; char *stack; bool done; };
struct Sched {
    ucontext_t main_ctx;
    std::vector<Fiber> fibers;
    Fiber *cur = nullptr;
    std::function<void()> *body = nullptr;
};
inline Sched &sched() { static thread_local Sched s; return s; }
inline double *shfl_buf() { static double buf[1024]; return buf; }
inline char *dyn_smem() { static char *p = (char *)aligned_alloc(64, 160 * 1024); return p; }
inline void fiber_entry() {
    Sched &S = sched();
    (*S.body)();
    S.cur->done = true;
    swapcontext(&S.cur->ctx, &S.main_ctx);
}
}  // namespace hipemu

extern thread_local uint3_emu threadIdx;
extern thread_local uint3_emu blockIdx;
extern thread_local dim3 blockDim;
extern thread_local dim3 gridDim;
#ifdef RM_HIPEMU_DEFINE_TLS
thread_local uint3_emu threadIdx;
thread_local uint3_emu blockIdx;
thread_local dim3 blockDim;
thread_local dim3 gridDim;
#endif

#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)hipemu::dyn_smem();

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline void __syncthreads() { hipemu::Sched &S = hipemu::sched(); swapcontext(&S.cur->ctx, &S.main_ctx); }
static const int warpSize = 64;

template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    static_assert(sizeof(T) <= sizeof(double), "shfl emu");
    std::memcpy(&buf[tid], &v, sizeof(T));
    __syncthreads();
    unsigned lane = tid % 64, src = lane + delta;
    T r = v;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    if ((lane % width) + delta < (unsigned)width && src < 64 && tid - lane + src < nthreads)
        std::memcpy(&r, &buf[tid - lane + src], sizeof(T));
    __syncthreads();
    return r;
}
template <typename T> inline T __shfl(T v, int src_lane, int width = 64) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    std::memcpy(&buf[tid], &v, sizeof(T));
    __syncthreads();
    unsigned lane = tid % 64, src = (unsigned)src_lane & 63u;
    T r = v;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    (void)width;
    if (tid - lane + src < nthreads) std::memcpy(&r, &buf[tid - lane + src], sizeof(T));
    __syncthreads();
    return r;
}
inline unsigned long long __ballot(int pred) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    buf[tid] = pred ? 1.0 : 0.0;
    __syncthreads();
    unsigned lane = tid % 64, nthreads = blockDim.x * blockDim.y * blockDim.z;
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; ++l)
        if (tid - lane + l < nthreads && buf[tid - lane + l] != 0.0) m |= 1ull << l;
    __syncthreads();
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
template <typename T> inline T __shfl_up(T v, unsigned delta, int width = 64) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    unsigned lane = tid % 64;
    (void)width;
    return __shfl(v, lane >= delta ? (int)(lane - delta) : (int)lane);
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    std::memcpy(&buf[tid], &v, sizeof(T));
    __syncthreads();
    unsigned lane = tid % 64, src = lane ^ (unsigned)mask;
    T r = v;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    (void)width;
    if (src < 64 && tid - lane + src < nthreads) std::memcpy(&r, &buf[tid - lane + src], sizeof(T));
    __syncthreads();
    return r;
}

inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) {
    auto *a = reinterpret_cast<std::atomic<unsigned long long> *>(p);
    unsigned long long o = a->load();
    while (o < v && !a->compare_exchange_weak(o, v)) {}
    return o;
}
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
    auto *a = reinterpret_cast<std::atomic<unsigned long long> *>(p);
    unsigned long long o = a->load();
    while (o > v && !a->compare_exchange_weak(o, v)) {}
    return o;
}
inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) {
    return reinterpret_cast<std::atomic<unsigned long long> *>(p)->fetch_or(v);
}
inline int atomicMin(int *p, int v) {
    auto *a = reinterpret_cast<std::atomic<int> *>(p);
    int cur = a->load();
    while (v < cur && !a->compare_exchange_weak(cur, v)) {}
    return cur;
}
inline int atomicCAS(int *p, int expected, int desired) {
    reinterpret_cast<std::atomic<int> *>(p)->compare_exchange_strong(expected, desired);
    return expected;   // the value found there (== the caller's `expected` when the swap happened)
}
inline int atomicAdd(int *p, int v) { return reinterpret_cast<std::atomic<int> *>(p)->fetch_add(v); }
inline unsigned atomicAdd(unsigned *p, unsigned v) { return reinterpret_cast<std::atomic<unsigned> *>(p)->fetch_add(v); }
inline unsigned atomicExch(unsigned *p, unsigned v) { return reinterpret_cast<std::atomic<unsigned> *>(p)->exchange(v); }
inline unsigned atomicMax(unsigned *p, unsigned v) {
    auto *a = reinterpret_cast<std::atomic<unsigned> *>(p);
    unsigned o = a->load();
    while (o < v && !a->compare_exchange_weak(o, v)) {}
    return o;
}
inline int __float2int_rn(float f) { return (int)std::lrintf(f); }
inline int atomicMax(int *p, int v) {
    auto *a = reinterpret_cast<std::atomic<int> *>(p);
    int o = a->load();
    while (o < v && !a->compare_exchange_weak(o, v)) {}
    return o;
}
inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long l) { double r; std::memcpy(&r, &l, 8); return r; }
inline int __float_as_int(float f) { int r; std::memcpy(&r, &f, 4); return r; }
inline float __int_as_float(int i) { float r; std::memcpy(&r, &i, 4); return r; }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_system() { std::atomic_thread_fence(std::memory_order_seq_cst); }

/* half: only widening loads are needed */
struct __half { uint16_t bits; };
inline float __half2float(__half h) {
    uint32_t s = (h.bits >> 15) & 1, e = (h.bits >> 10) & 31, m = h.bits & 1023, out;
    if (e == 0) {
        if (m == 0) out = s << 31;
        else { int sh = 0; while (!(m & 1024)) { m <<= 1; ++sh; } m &= 1023; out = (s << 31) | ((127 - 15 - sh + 1) << 23) | (m << 13); }
    } else if (e == 31) out = (s << 31) | 0x7f800000u | (m << 13);
    else out = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
    float f; std::memcpy(&f, &out, 4); return f;
}

/* ---- runtime API subset ---- */
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipHostMallocDefault 0
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 2; }
inline hipError_t hipFree(void *p) { free(p); return 0; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return 0; }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipErrorNotReady 600
inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)2; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return 0; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (void *)1; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
template <typename F> inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }

namespace hipemu {
template <typename F> void run_grid(dim3 grid, dim3 block, F &&body) {
    // Blocks run one after another on the calling OS thread; the threads of a block are fibers that
    // the scheduler resumes round-robin, each running until its next __syncthreads() (a yield) or
    // its end -- which gives barrier semantics as long as all live threads reach the same barriers.
    const unsigned nthreads = block.x * block.y * block.z;
    const size_t STACK = 128 * 1024;
    Sched &S = sched();
    std::function<void()> fn = [&] { body(); };
    S.body = &fn;
    if (S.fibers.size() < nthreads) {
        size_t old = S.fibers.size();
        S.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; ++i) S.fibers[i].stack = (char *)aligned_alloc(64, STACK);
    }
    blockDim = block; gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                for (unsigned t = 0; t < nthreads; ++t) {
                    Fiber &f = S.fibers[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = &S.main_ctx;
                    f.done = false;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                unsigned live = nthreads;
                while (live) {
                    live = 0;
                    for (unsigned t = 0; t < nthreads; ++t) {
                        Fiber &f = S.fibers[t];
                        if (f.done) continue;
                        threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
                        S.cur = &f;
                        swapcontext(&S.main_ctx, &f.ctx);
                        if (!f.done) ++live;
                    }
                }
            }
}
}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid(dim3(grid), dim3(block), [&] { kernel(__VA_ARGS__); })

#endif
