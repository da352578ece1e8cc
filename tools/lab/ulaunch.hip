// ulaunch.hip -- what does a kernel cost on MI355X before it moves a byte? Durations from dispatch timestamps
// (hipExtLaunchKernelGGL) and back-to-back wall time per launch, for empty / tiny-load kernels at several grid sizes.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <chrono>
#define CHK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k_empty(unsigned* out) { if (threadIdx.x == 9999) out[0] = 1; }
__global__ void k_store(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x; }
// each wave reads `n` KiB (n x 1 KiB wave loads, all issued up front), xor-reduces, lane 0 stores
template <int N>
__global__ void k_stream(const u4* __restrict__ src, unsigned* out, size_t stride_u4) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const u4* p = src + wave * stride_u4 + lane;
    u4 v[N];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = __builtin_nontemporal_load(p + i * 64);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) out[wave] = acc;
}
template <typename F>
int timeit(const char* name, F launch, int iters, size_t bytes) {
    std::vector<hipEvent_t> ev(2 * iters);
    for (auto& e : ev) CHK(hipEventCreate(&e));
    for (int w = 0; w < 5; w++) launch(nullptr, nullptr);
    CHK(hipDeviceSynchronize());
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < iters; i++) launch(ev[2 * i], ev[2 * i + 1]);
    CHK(hipDeviceSynchronize());
    auto t1 = std::chrono::high_resolution_clock::now();
    double tot = 0, mn = 1e9;
    for (int i = 0; i < iters; i++) { float ms; CHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms * 1e3; if (ms * 1e3 < mn) mn = ms * 1e3; }
    float span; CHK(hipEventElapsedTime(&span, ev[0], ev[2 * iters - 1]));
    double wall = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
    printf("%-34s kernel avg %7.2f us (min %6.2f) | start-to-end span/launch %7.2f us | host wall/launch %7.2f us", name, tot / iters, mn, span * 1e3 / iters, wall);
    if (bytes) printf(" | %7.1f GB/s (kernel) %7.1f GB/s (span)", bytes / (tot / iters) / 1e3, bytes / (span * 1e3 / iters) / 1e3);
    printf("\n");
    for (auto& e : ev) hipEventDestroy(e);
    return 0;
}
template <typename F>
int wallit(const char* name, F launch, int iters, size_t bytes) {
    for (int w = 0; w < 5; w++) launch();
    CHK(hipDeviceSynchronize());
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < iters; i++) launch();
    CHK(hipDeviceSynchronize());
    auto t1 = std::chrono::high_resolution_clock::now();
    double wall = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
    printf("PLAIN %-34s back-to-back %7.2f us per launch", name, wall);
    if (bytes) printf(" | %7.1f GB/s", bytes / wall / 1e3);
    printf("\n");
    return 0;
}
int main() {
    unsigned* out; CHK(hipMalloc(&out, 1 << 24));
    u4* src; size_t src_bytes = (size_t)3 << 30; CHK(hipMalloc(&src, src_bytes)); CHK(hipMemset(src, 1, src_bytes));
    hipStream_t s; CHK(hipStreamCreate(&s));
    int grids[] = {1, 256, 1024, 1376, 4096};
    for (int g : grids) {
        char nm[64]; snprintf(nm, 64, "empty  grid %d x256", g);
        timeit(nm, [&](hipEvent_t a, hipEvent_t b) { if (a) hipExtLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, a, b, 0, out); else hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, out); }, 200, 0);
    }
    timeit("store  grid 1376 x256", [&](hipEvent_t a, hipEvent_t b) { if (a) hipExtLaunchKernelGGL(k_store, dim3(1376), dim3(256), 0, s, a, b, 0, out); else hipLaunchKernelGGL(k_store, dim3(1376), dim3(256), 0, s, out); }, 200, 0);
    for (int g : grids) {
        char nm[64]; snprintf(nm, 64, "empty  grid %d x256", g);
        wallit(nm, [&]() { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, out); }, 2000, 0);
    }
    // streaming: rotate the base so nothing is cache resident (3 GiB ring)
    size_t off = 0;
#define STREAM(N, G, T)                                                                                                     \
    {                                                                                                                       \
        size_t per = (size_t)(G) * ((T) / 64) * (N) * 1024;                                                                 \
        char nm[64]; snprintf(nm, 64, "stream %d KiB/wave grid %d x%d (%.1f MB)", N, G, T, per / 1e6);                        \
        timeit(nm, [&](hipEvent_t a, hipEvent_t b) {                                                                        \
            if (off + per > src_bytes) off = 0;                                                                             \
            const u4* p = src + off / 16; off += per;                                                                       \
            if (a) hipExtLaunchKernelGGL((k_stream<N>), dim3(G), dim3(T), 0, s, a, b, 0, p, out, (size_t)(N) * 64);          \
            else hipLaunchKernelGGL((k_stream<N>), dim3(G), dim3(T), 0, s, p, out, (size_t)(N) * 64); }, 100, per);          \
        wallit(nm, [&]() {                                                                                                  \
            if (off + per > src_bytes) off = 0;                                                                             \
            const u4* p = src + off / 16; off += per;                                                                       \
            hipLaunchKernelGGL((k_stream<N>), dim3(G), dim3(T), 0, s, p, out, (size_t)(N) * 64); }, 400, per);               \
    }
    STREAM(8, 256, 256) STREAM(8, 1024, 256) STREAM(8, 1376, 256) STREAM(8, 2752, 256) STREAM(16, 1376, 256) STREAM(8, 5504, 256)
    STREAM(16, 2752, 256) STREAM(8, 688, 512) STREAM(32, 688, 256) STREAM(8, 16000, 256) STREAM(16, 8000, 256) STREAM(2, 5504, 256) STREAM(4, 2752, 256)
    // Infinity Cache (MALL) residency: re-stream the SAME buffer every launch (fits the 256 MiB LLC) vs the rotating ring
#define SAME(N, G, T)                                                                                                       \
    {                                                                                                                       \
        size_t per = (size_t)(G) * ((T) / 64) * (N) * 1024;                                                                 \
        char nm[64]; snprintf(nm, 64, "SAME-BUFFER %d KiB/wave grid %d (%.1f MB)", N, G, per / 1e6);                          \
        timeit(nm, [&](hipEvent_t a, hipEvent_t b) {                                                                        \
            if (a) hipExtLaunchKernelGGL((k_stream<N>), dim3(G), dim3(T), 0, s, a, b, 0, (const u4*)src, out, (size_t)(N) * 64); \
            else hipLaunchKernelGGL((k_stream<N>), dim3(G), dim3(T), 0, s, (const u4*)src, out, (size_t)(N) * 64); }, 100, per); \
    }
    SAME(8, 1376, 256) SAME(8, 2752, 256) SAME(8, 5504, 256) SAME(8, 256, 256)
    return 0;
}
