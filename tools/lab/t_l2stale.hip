// t_l2stale.hip -- which load flavours return a line another XCD rewrote (sc1 store) INSIDE the launch, when this XCD's L2 (and this CU's L1) already
// hold the old line? Decides whether a hand-off vector can be gathered with L2-cached loads (one fabric fetch per line and XCD) instead of sc1 loads
// (one per CU).   hipcc --offload-arch=gfx950 -O2 -o tools/lab/t_l2stale tools/lab/t_l2stale.hip && tools/lab/t_l2stale
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 64   // lines of 128 B tested
__device__ __forceinline__ unsigned ld_plain(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ unsigned ld_sc0sc1(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ unsigned ld_nt(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory"); }
__device__ void wait_flag(unsigned* f, unsigned want) { while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(4); }
__device__ void set_flag(unsigned* f, unsigned v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// roles by block: 0 = primer + reader on XCD 0; 8 = second reader on XCD 0 (another CU, never touched the lines); 1 = writer on XCD 1
__global__ void k(unsigned* data, unsigned* flags, unsigned* out, unsigned gen) {
    const unsigned b = blockIdx.x, l = threadIdx.x;   // 64 threads: one line each
    unsigned* line = data + l * 32;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (l == 0) out[1024 + b] = xcc & 15u;
    if (b == 0) {
        const unsigned v0 = ld_plain(line);                 // primes L1 (this CU) and L2 (XCD 0) with the old line
        __syncthreads();
        if (l == 0) set_flag(flags + 0, gen);
        if (l == 0) wait_flag(flags + 1, gen);
        __syncthreads();
        out[0 * N + l] = v0;
        // every line is touched by ONE flavour only (first touch after the rewrite): 16 lines each
        unsigned v;
        if (l < 16) v = ld_sc1(line); else if (l < 32) v = ld_nt(line); else if (l < 48) v = ld_plain(line); else v = ld_sc0sc1(line);
        out[1 * N + l] = v;
    } else if (b == 8 || b == 16 || b == 24 || b == 32) {   // other CUs of XCD 0: L1 cold, L2 holds the old line; one flavour per block
        if (l == 0) wait_flag(flags + 1, gen);
        __syncthreads();
        unsigned v;
        if (b == 8) v = ld_nt(line); else if (b == 16) v = ld_plain(line); else if (b == 24) v = ld_sc1(line); else v = ld_sc0sc1(line);
        out[(1 + b / 8) * N + l] = v;
    } else if (b == 1) {
        if (l == 0) wait_flag(flags + 0, gen);
        __syncthreads();
        st_sc1(line, gen);                                  // write-through store from XCD 1, acknowledged
        __syncthreads();
        if (l == 0) set_flag(flags + 1, gen);
    } else if (b == 9) {                                    // another CU on XCD 1 after the store: its L2 dropped the line (sc1 store)
        if (l == 0) wait_flag(flags + 1, gen);
        __syncthreads();
        out[6 * N + l] = ld_plain(line);
    } else if (b == 40) {                                   // XCD 0, after everything: plain loads behind an agent-scope acquire
        if (l == 0) wait_flag(flags + 1, gen);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        out[7 * N + l] = ld_plain(line);
    }
}
int main() {
    unsigned *data, *flags, *out;
    hipMalloc(&data, N * 128); hipMalloc(&flags, 256); hipMalloc(&out, 4096 * 4);
    hipMemset(data, 0, N * 128); hipMemset(flags, 0, 256);
    const char* names[8] = {"primer plain (old)", "same CU: 16 sc1|16 nt|16 plain|16 sc0sc1", "other CU nt", "other CU plain", "other CU sc1", "other CU sc0 sc1", "writer XCD other CU plain", "other CU plain after acquire"};
    for (unsigned gen = 1; gen <= 3; gen++) {
        hipMemset(out, 0xff, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(48), dim3(64), 0, 0, data, flags, out, gen);
        hipDeviceSynchronize();
        unsigned h[4096];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("gen %u  XCC of blocks 0,1,8,9: %u %u %u %u\n", gen, h[1024], h[1025], h[1032], h[1033]);
        for (int r = 0; r < 8; r++) {
            int fresh = 0, old = 0, other = 0;
            for (int i = 0; i < N; i++) { unsigned v = h[r * N + i]; if (v == gen) fresh++; else if (v == gen - 1) old++; else other++; }
            printf("  %-44s fresh %2d  old %2d  other %2d", names[r], fresh, old, other);
            if (r == 1) { printf("   per flavour fresh:"); for (int q = 0; q < 4; q++) { int f = 0; for (int i = 0; i < 16; i++) f += h[N + q * 16 + i] == gen; printf(" %d", f); } }
            printf("\n");
        }
    }
    return 0;
}
