// bench_micro/valu_rate.hip -- developer microbenchmark: issue cost (cycles per wave64 instruction on one SIMD) of the VALU
// instructions the narrow-dtype pyrDown chain is made of.  W waves per SIMD, 16 independent instructions per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP> __global__ __launch_bounds__(64) void k_rate(int iters, double *out, long long *cycles)
{
    double d[16]; float f[16]; unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.0f + i + threadIdx.x; u[i] = threadIdx.x * 2654435761u + i; }
    const double c6 = 6.0 + out[0] * 0;
    __shared__ double lut[256];
    for (int i = threadIdx.x; i < 256; i += 64) lut[i] = i * (1.0 / 255);
    __syncthreads();
    unsigned addr[16];
    const unsigned long long smask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(0x55555555 + (int)out[0]) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane(0x33333333 + (int)out[0]);
    const unsigned three = 3;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned r = (threadIdx.x * 2654435761u + i * 40503u) >> 7;
        // 23: every lane its own random entry; 24: camera-like (lanes within +-3 levels of each other); 25: all lanes one entry
        addr[i] = (OP == 23 || OP == 26) ? (r & 255) * 8 : OP == 24 ? ((128 + i + (r % 7)) & 255) * 8 : (i * 8);
    }
    asm volatile("s_mov_b64 s[20:21], 0x33333333\n s_mov_b64 vcc, 0x55555555" ::: "s20", "s21", "vcc");
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c6));
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c6));
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(c6));
#define CVTU(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
#define CVTF(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define CVTH(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(u[i]));
#define CVTB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));
#define BFE(i) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define DPPS(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define DPPR(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 15]));
#define MOV32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, -8" : "+v"(d[i]));
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "s20", "s21");
#define CNDV(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : );
#define CNDV64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : );
#define CMPCND(i) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "vcc");
#define CMPCNDS(i) asm volatile("v_cmp_lt_u32_e64 s[22:23], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "s22", "s23");
#define CMPONLY(i) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) & 15]) : "vcc");
#define CMPCNDF(i) asm volatile("v_cmp_lt_f64_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc" : : "v"(d[i]), "v"(d[(i + 1) & 15]), "v"(u[i]), "v"(u[(i + 1) & 15]) : "vcc");
#define MINF64(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
#define SDWASHL(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(u[i]) : "v"(three), "v"(u[(i + 1) & 15]));
#define LDSRD(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(addr[i]) : "memory");
#define DEPADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(c6));
        if (OP == 0) { REP16(ADD) } else if (OP == 1) { REP16(MUL) } else if (OP == 2) { REP16(FMA) } else if (OP == 3) { REP16(CVTU) }
        else if (OP == 4) { REP16(CVTF) } else if (OP == 5) { REP16(CVTH) } else if (OP == 6) { REP16(CVTB) } else if (OP == 7) { REP16(BFE) }
        else if (OP == 8) { REP16(DPPS) } else if (OP == 9) { REP16(DPPR) } else if (OP == 10) { REP16(CND) } else if (OP == 11) { REP16(MOV64) }
        else if (OP == 12) { REP16(MOV32) } else if (OP == 13) { REP16(LDEXP) } else if (OP == 14) { REP16(FMA32) } else if (OP == 15) { REP16(PERM) }
        else if (OP == 16) { REP16(LSHLADD) } else if (OP == 17) { REP16(CNDS) } else if (OP == 18) { REP16(CNDV) } else if (OP == 19) { REP16(PKADD) }
        else if (OP == 20) { REP16(PKFMA) } else if (OP == 21) { REP16(PKMUL) } else if (OP == 22) { REP16(SDWASHL) }
        else if (OP == 23 || OP == 24 || OP == 25) { REP16(LDSRD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if (OP == 26) { REP16(LDSRD) REP16(ADD) REP16(MUL) REP16(FMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if (OP == 27) { REP16(DEPADD) } else if (OP == 28) { REP16(CNDV64) } else if (OP == 29) { REP16(CMPCND) } else if (OP == 30) { REP16(CMPCNDS) }
        else if (OP == 31) { REP16(CMPONLY) } else if (OP == 32) { REP16(CMPCNDF) } else if (OP == 33) { REP16(MINF64) }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; float sf = 0; unsigned su = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s += d[i]; sf += f[i]; su += u[i]; }
    if (s + sf + su == 12345.678) out[1] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int OP> void run(const char *name)
{
    double *out; long long *cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 8)); CK(hipMemset(out, 0, 64));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int grid = 256 * 4 * wps;   // wps waves per SIMD (one wave per workgroup)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(64), 0, 0, 100, out, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(64), 0, 0, iters, out, cyc);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        // wall-clock cycles at 2.4 GHz per instruction issued on a SIMD (all of its waves together)
        printf("%-28s waves/SIMD %d : %.2f ns/instr/SIMD = %.2f cyc@2.4GHz (counter: %.2f ticks per instr of one wave)\n", name, wps,
               ms * 1e6 / ((double)iters * 16 * wps), ms * 1e6 / ((double)iters * 16 * wps) * 2.4, (double)c / ((double)iters * 16));
    }
}

int main()
{
    run<0>("v_add_f64"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<3>("v_cvt_f64_u32"); run<4>("v_cvt_f64_f32"); run<5>("v_cvt_f32_f16");
    run<6>("v_cvt_f32_ubyte1"); run<7>("v_bfe_u32"); run<8>("v_mov_b32_dpp wave_shr:1"); run<9>("v_mov_b32_dpp row_shr:1"); run<10>("v_cndmask_b32");
    run<11>("v_mov_b64"); run<12>("v_mov_b32"); run<13>("v_ldexp_f64"); run<14>("v_fma_f32"); run<15>("v_perm_b32"); run<16>("v_lshl_add_u32");
    run<17>("v_cndmask_b32_e64 sgpr mask"); run<18>("v_cndmask_b32_e32 vcc"); run<19>("v_pk_add_f32"); run<20>("v_pk_fma_f32"); run<21>("v_pk_mul_f32");
    run<22>("v_lshlrev_b32_sdwa BYTE_1"); run<23>("ds_read_b64 LUT random"); run<24>("ds_read_b64 LUT camera-like"); run<25>("ds_read_b64 LUT uniform");
    run<26>("16 ds_read_b64 + 48 f64 VALU (cost per 16 = x16)"); run<27>("v_add_f64 dependent chain");
    run<28>("v_cndmask_b32_e64 vcc"); run<29>("v_cmp_e32 vcc + v_cndmask_e32 (pair)"); run<30>("v_cmp_e64 sgpr + v_cndmask_e64 (pair)"); run<31>("v_cmp_lt_u32_e32 vcc");
    run<32>("v_cmp_lt_f64 + v_cndmask (pair)"); run<33>("v_min_f64");
    return 0;
}
