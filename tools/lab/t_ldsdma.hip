// t_ldsdma.hip -- where does `buffer_load_dwordx4 ... lds` put its bytes on gfx950? (tools/, not product)
// Probes, each on a poisoned 160 KiB LDS: M0 below and above 64 KiB, the instruction's immediate offset (does it move the
// LDS address as well as the global one?), the 4-byte form, and vmcnt accounting. Prints where the data landed.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lab/t_ldsdma tools/lab/t_ldsdma.hip && tools/lab/t_ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

constexpr unsigned LDS_BYTES = 160 * 1024;
constexpr unsigned POISON = 0xDEADBEEFu;

template <int VARIANT>
__global__ void __launch_bounds__(64) probe(const unsigned* src, unsigned nbytes, unsigned m0v, unsigned* report) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* l = reinterpret_cast<unsigned*>(smem);
    for (unsigned i = threadIdx.x; i < LDS_BYTES / 4; i += 64) l[i] = POISON;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const unsigned voff = threadIdx.x * (VARIANT == 3 ? 4 : 16);
    const unsigned soff = 4096;   // global byte offset of the piece
    if (VARIANT == 0)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
    if (VARIANT == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 nt lds" ::"s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
    if (VARIANT == 3)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report: first and last non-poison dword index, count, value at first, value at first + 4 (next lane's first dword)
    if (threadIdx.x == 0) {
        unsigned first = ~0u, last = 0, cnt = 0;
        for (unsigned i = 0; i < LDS_BYTES / 4; i++)
            if (l[i] != POISON) { if (first == ~0u) first = i; last = i; cnt++; }
        report[0] = first; report[1] = last; report[2] = cnt;
        report[3] = first != ~0u ? l[first] : 0; report[4] = first != ~0u ? l[first + 1] : 0;
        report[5] = first != ~0u ? l[first + 4] : 0; report[6] = first != ~0u ? l[last] : 0;
    }
}

int main() {
    const unsigned n = 1 << 16;   // dwords; src[i] = i
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; i++) h[i] = i;
    unsigned *src, *rep;
    hipMalloc(&src, n * 4); hipMalloc(&rep, 64);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    struct { int variant; unsigned m0; const char* what; } cases[] = {
        {0, 8192, "dwordx4, m0 = 8 KiB, no immediate"},
        {1, 8192, "dwordx4, m0 = 8 KiB, offset:1024"},
        {0, 96 * 1024, "dwordx4, m0 = 96 KiB, no immediate"},
        {0, 150 * 1024, "dwordx4, m0 = 150 KiB, no immediate"},
        {3, 8192, "dword, m0 = 8 KiB"},
        {3, 150 * 1024, "dword, m0 = 150 KiB"},
    };
    for (auto& c : cases) {
        hipMemset(rep, 0, 64);
        if (c.variant == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), LDS_BYTES, 0, src, n * 4, c.m0, rep);
        if (c.variant == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), LDS_BYTES, 0, src, n * 4, c.m0, rep);
        if (c.variant == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), LDS_BYTES, 0, src, n * 4, c.m0, rep);
        unsigned r[8];
        hipError_t e = hipMemcpy(r, rep, 32, hipMemcpyDeviceToHost);
        printf("%-40s rc=%d: LDS bytes [%u, %u] touched dwords %u; first dword holds src[%u], next %u, +16B %u, last %u\n", c.what, (int)e,
               r[0] * 4, r[1] * 4 + 3, r[2], r[3], r[4], r[5], r[6]);
    }
    return 0;
}
