// v_mfma_f32_4x4x4_16b_f16 as a per-lane dot4 for the int4 GEMV: (1) are fp16 DENORMAL A inputs multiplied exactly?
// (2) issue cost next to the nibble-extraction VALU ops, against the v_dot2c form.
// hipcc --offload-arch=gfx950 -O2 tools/lab/t_mfma.hip -o tools/lab/t_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__global__ void exact(const unsigned* w, const u2* x, float* out) {
    const int l = threadIdx.x;
    const unsigned ww = w[l];
    u2 a = {ww & 0x000F000Fu, (ww >> 8) & 0x000F000Fu};           // (n0,n4),(n2,n6) * 2^-24
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, x[l]), d, 0, 0, 0);
    out[l] = d[l & 3];
}

template <int MODE>
__global__ void rate(const unsigned* w, const u2* x, float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    unsigned ww = w[l];
    u2 xa = x[l], xb = x[(l + 7) & 63];
    f4 de = {0, 0, 0, 0}, dq = {0, 0, 0, 0};
    float ae = 0.f, ao = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned tt = ww >> 8;
            const unsigned e0 = ww & 0x000F000Fu, o0 = ww & 0x00F000F0u, e1 = tt & 0x000F000Fu, o1 = tt & 0x00F000F0u;
            if (MODE == 0) {
                ae = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, e0), __builtin_bit_cast(h2, xa[0]), ae, false);
                ao = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, o0), __builtin_bit_cast(h2, xb[0]), ao, false);
                ae = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, e1), __builtin_bit_cast(h2, xa[1]), ae, false);
                ao = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, o1), __builtin_bit_cast(h2, xb[1]), ao, false);
            } else {
                u2 A = {e0, e1}, B = {o0, o1};
                de = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4, A), __builtin_bit_cast(h4, xa), de, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4, B), __builtin_bit_cast(h4, xb), dq, 0, 0, 0);
            }
            ww = ww * 1664525u + 1013904223u + u;      // next "weight" dword (2 VALU, stands for the load data)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ae + ao + de[l & 3] + dq[l & 3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<unsigned> hw(64);
    std::vector<uint16_t> hx(64 * 4);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& v : hw) v = rnd() * 977u + rnd();
    for (auto& v : hx) { float f = ((int)(rnd() % 2001) - 1000) / 500.0f; _Float16 h = (_Float16)f; v = *(uint16_t*)&h; }
    unsigned* dw; u2* dx; float* dout; unsigned long long* dc;
    hipMalloc(&dw, 256); hipMalloc(&dx, 512); hipMalloc(&dout, 4 << 20); hipMalloc(&dc, 8 * 4096);
    hipMemcpy(dw, hw.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact, dim3(1), dim3(64), 0, 0, dw, dx, dout);
    float ho[64]; hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
    double worst = 0; int zeros = 0;
    for (int l = 0; l < 64; l++) {
        const unsigned ww = hw[l];
        const int q[4] = {(int)(ww & 15), (int)((ww >> 16) & 15), (int)((ww >> 8) & 15), (int)((ww >> 24) & 15)};
        double ref = 0;
        for (int k = 0; k < 4; k++) { _Float16 h = *(_Float16*)&hx[l * 4 + k]; ref += q[k] * (double)(float)h; }
        ref /= 16777216.0;
        if (ho[l] == 0.f && ref != 0) zeros++;
        worst = fmax(worst, fabs(ho[l] - ref) / fmax(1e-12, fabs(ref)));
    }
    printf("mfma 4x4x4 f16 with denormal A: worst rel err %.3g, flushed-to-zero lanes %d/64 (sample %.9g)\n", worst, zeros, ho[5]);
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 8; waves++) {
        for (int mode = 0; mode < 2; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * waves), dim3(256), 0, 0, dw, dx, dout, dc, iters);
                else hipLaunchKernelGGL(rate<1>, dim3(256 * waves), dim3(256), 0, 0, dw, dx, dout, dc, iters);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned long long hc[256]; hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : hc) avg += c; avg /= 256;
            printf("%d waves/SIMD %-6s: wave sees %.1f cycles per weight dword; kernel %.1f us -> %.2f ns per wave-dword per SIMD\n", waves, mode ? "mfma" : "dot2c",
                   avg / (iters * 8.0), ms * 1e3, ms * 1e6 / (iters * 8.0) / waves);
        }
    }
    return 0;
}
