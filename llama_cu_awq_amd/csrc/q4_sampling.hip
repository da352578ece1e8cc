// q4_sampling.hip -- temperature / top-p sampling: sample() sampler.h:51-81 with softmax_logits_kernel
// (gpu_kernels.h:499-550) and sample_top_p_kernel (:555-584). The reference leans on cub::DeviceRadixSort and
// cub::DeviceScan; here:
//   * softmax_logits: one 1024-thread block, the same fp16 rounding points (logits/T, exp, normalise); sums are a
//     balanced pairwise tree over the 1024 per-thread partials (cub's order is unspecified; the oracle restates this one)
//   * sort: single-block stable LSD radix sort of the fp16 probabilities (non-negative -> bit pattern is monotonic),
//     4 passes of 4 bits, descending, ties keep ascending index (what cub's stable SortPairsDescending yields)
//   * prefix sum + threshold search fused: the reference's inclusive scan accumulates IN FP16 and the search wants the
//     first index whose prefix reaches the threshold, so one lane walks the sorted probabilities with an fp16
//     accumulator and stops at the hit (a few dozen steps for top-p 0.6-0.9; vocab steps worst case).
#include <hip/hip_runtime.h>
#include "q4_device.h"
#include "q4_internal.h"
using namespace q4;

namespace {

__device__ __forceinline__ float block_tree_sum(float v, float* red) {   // red: 16 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return row16_sum(red[lane & 15]);
}
__device__ __forceinline__ float block_tree_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return row16_max(red[lane & 15]);
}

__global__ void __launch_bounds__(1024) softmax_logits_kernel(q4_half* logits, int size, float temperature, int* indices) {
    __shared__ float red[16];
    const int tid = threadIdx.x, step = blockDim.x;
    for (int t = tid; t < size; t += step) {
        indices[t] = t;                                        // gpu_kernels.h:507
        float val = h2f(logits[t]);
        val /= temperature;                                    // :511
        logits[t] = f2h(val);
    }
    __syncthreads();
    float max_val = tid < size ? h2f(logits[tid]) : -3.402823466e+38f;   // :522
    for (int i = tid + step; i < size; i += step) max_val = fmaxf(max_val, h2f(logits[i]));
    max_val = block_tree_max(max_val, red);
    float sum = 0.0f;
    for (int i = tid; i < size; i += step) {
        const float v = expf(h2f(logits[i]) - max_val);        // :536
        logits[i] = f2h(v);
        sum += v;
    }
    sum = block_tree_sum(sum, red);
    for (int t = tid; t < size; t += step) logits[t] = f2h(h2f(logits[t]) / sum);   // :549
}

// one pass of the stable LSD radix sort: 4-bit digit at `shift`, descending (digit' = 15 - digit)
__global__ void __launch_bounds__(1024) radix_pass_kernel(const uint16_t* kin, const int* vin, uint16_t* kout, int* vout, int n,
                                                          int shift) {
    __shared__ unsigned cnt[16 * 1024];       // [digit][thread]
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + 1023) / 1024;
    const int lo = tid * chunk, hi = min(n, lo + chunk);
    unsigned c[16];
#pragma unroll
    for (int d = 0; d < 16; d++) c[d] = 0;
    for (int i = lo; i < hi; i++) {
        const int d = 15 - ((kin[i] >> shift) & 15);
#pragma unroll
        for (int e = 0; e < 16; e++) c[e] += (e == d);
    }
#pragma unroll
    for (int d = 0; d < 16; d++) cnt[d * 1024 + tid] = c[d];
    __syncthreads();
    // exclusive scan over the flattened [digit][thread] array: thread t owns entries 16t .. 16t+15
    unsigned local[16], total = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) { local[e] = cnt[tid * 16 + e]; total += local[e]; }
    unsigned incl = total;                                   // wave inclusive scan
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    unsigned run = base + incl - total;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; e++) { cnt[tid * 16 + e] = run; run += local[e]; }
    __syncthreads();
    unsigned off[16];
#pragma unroll
    for (int d = 0; d < 16; d++) off[d] = cnt[d * 1024 + tid];
    for (int i = lo; i < hi; i++) {
        const uint16_t k = kin[i];
        const int d = 15 - ((k >> shift) & 15);
        unsigned dst = 0;
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (e == d) dst = off[e]++;
        kout[dst] = k;
        vout[dst] = vin[i];
    }
}

// sample_top_p_kernel gpu_kernels.h:555-584 fused with the fp16 inclusive scan (sampler.h:72-78)
__global__ void sample_scan_kernel(const uint16_t* probs, const int* indices, int n, float threshold, int* result,
                                   volatile int* pPos, int* pPosGpu) {
    if (threadIdx.x != 0) return;
    float run = 0.f;
    int min_index = n - 1;
    for (int t = 0; t < n; t++) {
        run = round_h(run + h2f(probs[t]));              // fp16 accumulator
        if (run >= threshold) { min_index = t; break; }
    }
    int token_pos = *pPosGpu;                            // == *pPos (:579-580) without the PCIe read
    token_pos++;
    result[token_pos] = indices[min_index];              // :578
    __threadfence_system();                              // the host may be spinning on *pPos (q4_wait_pos)
    *pPos = token_pos;
    *pPosGpu = token_pos;
}

}  // namespace

extern "C" int q4_sample_topp_device(Sampler* sampler, RunState* s, float coin) {
    const int n = sampler->vocab_size;
    Q4_LAUNCH(softmax_logits_kernel, dim3(1), dim3(1024), 0, s->logits, n, sampler->temperature, sampler->indices);   // sampler.h:53
    Q4_LAUNCH_CHECK();
    float threshold;
    const uint16_t* keys = s->logits;
    const int* vals = sampler->indices;
    if (sampler->topp <= 0 || sampler->topp >= 1) {
        threshold = coin;                                                           // sampler.h:57-59
    } else {
        if (sampler->temp_storage_bytes_sort == 0) {                                // :62-66 (lazy scratch)
            const size_t bytes = 2 * ((size_t)n * sizeof(uint16_t) + 256) + 2 * ((size_t)n * sizeof(int) + 256);
            Q4_HIP(hipMalloc(&sampler->tempStorage_sort, bytes));
            sampler->temp_storage_bytes_sort = bytes;
        }
        char* base = (char*)sampler->tempStorage_sort;
        const size_t kb = ((size_t)n * sizeof(uint16_t) + 255) / 256 * 256, vb = ((size_t)n * sizeof(int) + 255) / 256 * 256;
        uint16_t* k0 = (uint16_t*)base; uint16_t* k1 = (uint16_t*)(base + kb);
        int* v0 = (int*)(base + 2 * kb); int* v1 = (int*)(base + 2 * kb + vb);
        Q4_LAUNCH(radix_pass_kernel, dim3(1), dim3(1024), 0, (const uint16_t*)s->logits, (const int*)sampler->indices, k0, v0, n, 0);
        Q4_LAUNCH(radix_pass_kernel, dim3(1), dim3(1024), 0, (const uint16_t*)k0, (const int*)v0, k1, v1, n, 4);
        Q4_LAUNCH(radix_pass_kernel, dim3(1), dim3(1024), 0, (const uint16_t*)k1, (const int*)v1, k0, v0, n, 8);
        Q4_LAUNCH(radix_pass_kernel, dim3(1), dim3(1024), 0, (const uint16_t*)k0, (const int*)v0, k1, v1, n, 12);
        Q4_LAUNCH_CHECK();
        keys = k1;
        vals = v1;
        threshold = coin * sampler->topp;                                           // sampler.h:69
    }
    Q4_LAUNCH(sample_scan_kernel, dim3(1), dim3(64), 0, keys, vals, n, threshold, &(s->shared_data->tokens[0]),
              &(s->shared_data->pos), s->pos);                                      // :80
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
