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
 * fibers (a 20-line x86-64 stack switch: swapcontext's two signal-mask system calls per yield were most of
 * the CPU suite's run time) resumed round-robin.  Cross-lane operations of a WAVE (shuffles, DPP, ballots, MFMA,
 * wave barriers) are a write, a yield, a read, a yield: every lane of the wave runs the same sequence, so one
 * round of the scheduler separates the phases.  __syncthreads() / s_barrier park a fiber until every LIVE fiber
 * of the block has arrived (waves may reach it after different numbers of wave-level yields; a finished wave
 * no longer takes part); __shared__ variables are
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
struct Fiber { void *sp; char *stack; bool done; bool at_barrier; };
struct Sched {
    void *main_sp = nullptr;
    std::vector<Fiber> fibers;
    Fiber *cur = nullptr;
    std::function<void()> *body = nullptr;
};
inline Sched &sched() { static thread_local Sched s; return s; }
inline double *shfl_buf() { static double buf[1024]; return buf; }
inline char *dyn_smem() { static char *p = (char *)aligned_alloc(64, 160 * 1024); return p; }
}  // namespace hipemu
// saves the callee-saved registers and the stack pointer of the running context in *save_sp, continues the context whose stack
// pointer is load_sp (System V x86-64: rbx, rbp, r12-r15; the floating-point control words never change here)
extern "C" void hipemu_swap(void **save_sp, void *load_sp);
#ifdef RM_HIPEMU_DEFINE_TLS
#if !defined(__x86_64__)
#error "the host emulation's fiber switch is written for x86-64"
#endif
asm(".text\n.weak hipemu_swap\n.type hipemu_swap,@function\nhipemu_swap:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size hipemu_swap, .-hipemu_swap\n");
#endif
namespace hipemu {
inline void fiber_entry() {
    Sched &S = sched();
    (*S.body)();
    S.cur->done = true;
    hipemu_swap(&S.cur->sp, S.main_sp);
    abort();   // a finished fiber is never resumed
}
// a fresh fiber: its first resume pops six zeroed registers and returns into fiber_entry with the stack the ABI expects at a function's
// first instruction (16-byte aligned before the return address was pushed)
inline void fiber_init(Fiber &f, size_t stack_bytes) {
    uintptr_t top = ((uintptr_t)f.stack + stack_bytes) & ~(uintptr_t)15;
    void **sp = (void **)(top - 16);          // [sp + 8] unused pad, [sp] = return address slot -> after `ret` rsp = top - 8 (== 8 mod 16)
    *sp = (void *)&fiber_entry;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = (void *)sp;
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
namespace hipemu { inline void yield() { Sched &S = sched(); hipemu_swap(&S.cur->sp, S.main_sp); } }
inline void __syncthreads() { hipemu::Sched &S = hipemu::sched(); S.cur->at_barrier = true; hipemu_swap(&S.cur->sp, S.main_sp); }
static const int warpSize = 64;

template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    static_assert(sizeof(T) <= sizeof(double), "shfl emu");
    std::memcpy(&buf[tid], &v, sizeof(T));
    hipemu::yield();
    unsigned lane = tid % 64, src = lane + delta;
    T r = v;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    if ((lane % width) + delta < (unsigned)width && src < 64 && tid - lane + src < nthreads)
        std::memcpy(&r, &buf[tid - lane + src], sizeof(T));
    hipemu::yield();
    return r;
}
template <typename T> inline T __shfl(T v, int src_lane, int width = 64) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    std::memcpy(&buf[tid], &v, sizeof(T));
    hipemu::yield();
    unsigned lane = tid % 64, src = (unsigned)src_lane & 63u;
    T r = v;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    (void)width;
    if (tid - lane + src < nthreads) std::memcpy(&r, &buf[tid - lane + src], sizeof(T));
    hipemu::yield();
    return r;
}
inline unsigned long long __ballot(int pred) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    buf[tid] = pred ? 1.0 : 0.0;
    hipemu::yield();
    unsigned lane = tid % 64, nthreads = blockDim.x * blockDim.y * blockDim.z;
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; ++l)
        if (tid - lane + l < nthreads && buf[tid - lane + l] != 0.0) m |= 1ull << l;
    hipemu::yield();
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
    hipemu::yield();
    unsigned lane = tid % 64, src = lane ^ (unsigned)mask;
    T r = v;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    (void)width;
    if (src < 64 && tid - lane + src < nthreads) std::memcpy(&r, &buf[tid - lane + src], sizeof(T));
    hipemu::yield();
    return r;
}

/* ---- gfx950 cross-lane intrinsics the product kernels use directly (so that the code that ships is the code the CPU suite runs) ----
 * DPP (data-parallel primitives): lane i reads the `src` of another lane chosen by dpp_ctrl; a lane whose row / bank is masked out,
 * or whose source does not exist (bound_ctrl false), keeps `old`.  Controls modelled: quad_perm (0x00-0xFF), row_shl / row_shr /
 * row_ror (0x101-0x12F), wave_shl:1 0x130, wave_rol:1 0x134, wave_shr:1 0x138, wave_ror:1 0x13C, row_mirror 0x140,
 * row_half_mirror 0x141, row_bcast:15 0x142, row_bcast:31 0x143 (CDNA ISA, "DPP_CTRL"). */
inline int hipemu_dpp_source(int lane, int ctrl) {
    const int row = lane & ~15, l = lane & 15;
    if (ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return l + n <= 15 ? lane + n : -1; }
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return l - n >= 0 ? lane - n : -1; }
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row | ((l - n) & 15); }
    if (ctrl == 0x130) return lane < 63 ? lane + 1 : -1;
    if (ctrl == 0x134) return (lane + 1) & 63;
    if (ctrl == 0x138) return lane > 0 ? lane - 1 : -1;
    if (ctrl == 0x13C) return (lane - 1) & 63;
    if (ctrl == 0x140) return row | (15 - l);
    if (ctrl == 0x141) return row | (l & 8) | (7 - (l & 7));
    if (ctrl == 0x142) return lane >= 16 ? row - 1 : -1;          /* lane 15 of the previous row */
    if (ctrl == 0x143) return lane >= 32 ? 31 : -1;               /* lane 31 into the upper half */
    fprintf(stderr, "hipemu: dpp_ctrl 0x%x is not modelled\n", ctrl); abort();
}
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double *buf = hipemu::shfl_buf();
    std::memcpy(&buf[tid], &src, sizeof(int));
    hipemu::yield();
    const int lane = (int)(tid % 64);
    int r = old;
    if (((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane >> 2) & 3)) & 1)) {
        const int sl = hipemu_dpp_source(lane, ctrl);
        const unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
        if (sl >= 0 && tid - lane + sl < nthreads) std::memcpy(&r, &buf[tid - lane + sl], sizeof(int));
        else if (bound_ctrl) r = 0;
    }
    hipemu::yield();
    return r;
}
inline int __double2loint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(unsigned)(b & 0xffffffffll); }
inline int __double2hiint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(unsigned)((unsigned long long)b >> 32); }
inline double __hiloint2double(int hi, int lo) { unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double d; std::memcpy(&d, &b, 8); return d; }
inline int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane); }
/* (used on wave-uniform values only -- the product's contract for putting them into scalar registers: the value itself) */
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
/* s_barrier: workgroup barrier without the memory waits of __syncthreads(); a finished wave no longer takes part (fiber model: a yield) */
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
/* wave-level ordering points: lanes are fibers here, so the wave barrier must really hand over; fences order nothing extra */
inline void __builtin_amdgcn_wave_barrier() { hipemu::yield(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
// v_alignbyte_b32: ({hi, lo} >> 8 * (shift & 3)) & 0xffffffff;  v_dot4_u32_u8: four byte products + c (no clamp used)
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned shift) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (shift & 3)));
}
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) {
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return c;
}
template <typename T> inline T __builtin_nontemporal_load(const T *p) { return *p; }
/* v_mfma_f64_16x16x4_f64: D[16x16] = A[16x4] B[4x16] + C.  Lane l holds A[l % 16][l / 16], B[l / 16][l % 16] and the four
 * elements D[4 r + l / 16][l % 16], r = 0..3 (CDNA3/4 ISA, "MFMA 16x16x4 F64").  The products of one instruction are accumulated
 * k = 0..3 with fused multiply-adds; the hardware's internal order is not documented, which is why the tests of the kernels that use
 * it compare against scipy within 1e-11 and never bit for bit. */
typedef double hipemu_v4f64 __attribute__((vector_size(32)));
inline hipemu_v4f64 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_v4f64 c, int, int, int) {
    static double abuf[1024], bbuf[1024];
    unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    abuf[tid] = a; bbuf[tid] = b;
    hipemu::yield();
    const unsigned lane = tid % 64, w0 = tid - lane, j = lane % 16;
    hipemu_v4f64 d = c;
    for (int r = 0; r < 4; ++r) {
        const unsigned i = 4 * r + lane / 16;
        double acc = c[r];
        for (unsigned k = 0; k < 4; ++k) acc = std::fma(abuf[w0 + k * 16 + i], bbuf[w0 + k * 16 + j], acc);
        d[r] = acc;
    }
    hipemu::yield();
    return d;
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
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return reinterpret_cast<std::atomic<unsigned long long> *>(p)->fetch_add(v); }
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
#define hipDeviceAttributeMultiprocessorCount 63
inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 256; return 0; }
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
                    f.done = false; f.at_barrier = false;
                    fiber_init(f, STACK);
                }
                unsigned live = nthreads;
                while (live) {
                    live = 0;
                    unsigned parked = 0;
                    for (unsigned t = 0; t < nthreads; ++t) {
                        Fiber &f = S.fibers[t];
                        if (f.done) continue;
                        if (f.at_barrier) { ++live; ++parked; continue; }   // waits for the rest of the block
                        threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
                        S.cur = &f;
                        hipemu_swap(&S.main_sp, f.sp);
                        if (!f.done) { ++live; if (f.at_barrier) ++parked; }
                    }
                    if (live && parked == live)   // every live fiber has arrived: the barrier opens
                        for (unsigned t = 0; t < nthreads; ++t) S.fibers[t].at_barrier = false;
                }
            }
}
}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid(dim3(grid), dim3(block), [&] { kernel(__VA_ARGS__); })

#endif
