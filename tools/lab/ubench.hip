// ubench.hip -- VALU issue-rate / dependent-latency microbenchmarks for the instructions the int4 GEMV leans on.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lab/ubench.hip -o tools/lab/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(NAME, INDEP, DEP)                                                                       \
    __global__ void NAME##_tput(unsigned* out, uint64_t* cyc, int iters, unsigned seed) {               \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        unsigned b = seed * 31 + threadIdx.x, c = seed * 37 + 1;                                       \
        uint64_t t0 = __builtin_readcyclecounter();                                                     \
        for (int i = 0; i < iters; i++) { BODY8(INDEP) BODY8(INDEP) BODY8(INDEP) BODY8(INDEP) }         \
        uint64_t t1 = __builtin_readcyclecounter();                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                      \
    }                                                                                                   \
    __global__ void NAME##_lat(unsigned* out, uint64_t* cyc, int iters, unsigned seed) {                \
        unsigned a0 = seed + threadIdx.x;                                                               \
        unsigned b = seed * 31 + threadIdx.x, c = seed * 37 + 1;                                        \
        uint64_t t0 = __builtin_readcyclecounter();                                                     \
        for (int i = 0; i < iters; i++) { DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP DEP } \
        uint64_t t1 = __builtin_readcyclecounter();                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0;                                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                      \
    }

#define I_FMA(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "v"(c));
#define I_PKFMA16(n) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "v"(c));
#define I_PKADD16(n) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define I_DOT2C(n) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_DOT2(n) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "v"(c));
#define I_ANDOR(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_LSHR(n) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a##n));
#define I_FMAMIX(n) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a##n) : "v"(b), "v"(c));
#define I_CVTUB(n) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a##n));
#define I_PKMUL16(n) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define I_PERM(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_BFE(n) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(a##n));
#define I_MADU24(n) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "v"(c));
#define I_ADDF(n) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define I_MULF(n) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define I_CVTF16(n) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a##n));

KERNEL(fma, I_FMA, I_FMA(0))
KERNEL(pkfma16, I_PKFMA16, I_PKFMA16(0))
KERNEL(pkadd16, I_PKADD16, I_PKADD16(0))
KERNEL(dot2c, I_DOT2C, I_DOT2C(0))
KERNEL(dot2, I_DOT2, I_DOT2(0))
KERNEL(andor, I_ANDOR, I_ANDOR(0))
KERNEL(lshr, I_LSHR, I_LSHR(0))
KERNEL(fmamix, I_FMAMIX, I_FMAMIX(0))
KERNEL(cvtub, I_CVTUB, I_CVTUB(0))
KERNEL(pkmul16, I_PKMUL16, I_PKMUL16(0))
KERNEL(perm, I_PERM, I_PERM(0))
KERNEL(bfe, I_BFE, I_BFE(0))
KERNEL(madu24, I_MADU24, I_MADU24(0))
KERNEL(addf, I_ADDF, I_ADDF(0))
KERNEL(mulf, I_MULF, I_MULF(0))
KERNEL(cvtf16, I_CVTF16, I_CVTF16(0))

typedef void (*kfn)(unsigned*, uint64_t*, int, unsigned);
struct Entry { const char* name; kfn tput; kfn lat; };

int main() {
    Entry es[] = {{"v_fma_f32", fma_tput, fma_lat}, {"v_pk_fma_f16", pkfma16_tput, pkfma16_lat}, {"v_pk_add_f16", pkadd16_tput, pkadd16_lat},
                  {"v_dot2c_f32_f16", dot2c_tput, dot2c_lat}, {"v_dot2_f32_f16", dot2_tput, dot2_lat}, {"v_and_or_b32", andor_tput, andor_lat},
                  {"v_lshrrev_b32", lshr_tput, lshr_lat}, {"v_fma_mix_f32", fmamix_tput, fmamix_lat}, {"v_cvt_f32_ubyte0", cvtub_tput, cvtub_lat},
                  {"v_pk_mul_f16", pkmul16_tput, pkmul16_lat}, {"v_perm_b32", perm_tput, perm_lat}, {"v_bfe_u32", bfe_tput, bfe_lat},
                  {"v_mad_u32_u24", madu24_tput, madu24_lat}, {"v_add_f32", addf_tput, addf_lat}, {"v_mul_f32", mulf_tput, mulf_lat},
                  {"v_cvt_f32_f16", cvtf16_tput, cvtf16_lat}};
    unsigned* out; uint64_t* cyc;
    CHK(hipMalloc(&out, 1024 * 2048 * 4)); CHK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-18s | ns per wave-instruction per SIMD at 1/2/4/8 waves per SIMD (wall clock) | s_memtime ticks/instr at 1w and 4w | dep-chain ns\n", "instr");
    for (auto& e : es) {
        double ns[5], ticks[5];
        int wps[4] = {1, 2, 4, 8};
        for (int k = 0; k < 5; k++) {
            int w = k < 4 ? wps[k] : 1;
            int threads = w >= 4 ? 1024 : 256 * w;          // waves per block
            int blocks = 256 * (w == 8 ? 2 : 1);            // 8 w/SIMD = 2 blocks of 16 waves per CU
            uint64_t h = 0;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k < 4 ? e.tput : e.lat, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 12345u + rep);
                CHK(hipEventRecord(e1, 0));
                CHK(hipDeviceSynchronize());
                CHK(hipEventElapsedTime(&ms, e0, e1));
                CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            }
            double instr_per_simd = 32.0 * iters * w;
            ns[k] = ms * 1e6 / instr_per_simd;
            ticks[k] = (double)h / (32.0 * iters);   // per wave
        }
        printf("%-18s | %6.3f %6.3f %6.3f %6.3f | %6.2f %6.2f | %6.3f\n", e.name, ns[0], ns[1], ns[2], ns[3], ticks[0], ticks[2], ns[4]);
    }
    return 0;
}
