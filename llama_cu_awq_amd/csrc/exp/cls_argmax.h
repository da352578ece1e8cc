// exp/cls_argmax.h -- LABORATORY (libllama2_q4_prof.so only, profiling knob 12). The greedy sampler (argmax_kernel, gpu_kernels.h:448-493;
// llama2_q4.cu:384 -> sampler.h:47-49) as the classifier launch's epilogue. Every wave keeps the
// best of its own rows (the fp16-rounded logit it stores, first maximum = lowest row), a block leaves ONE candidate {value, row} (written through,
// drained) and one returning arrival; the block that arrives last reads the 256 candidates, decides with argmax_kernel's rule (value, then the lower
// index), copies the winner's embedding row for the next step where asked, writes the token ring and advances both position words. One launch and one
// boundary fewer per greedy token; the logits are stored as before. Bit-identical, measured level (7B and 13B -n 256, interleaved in one process, three
// calls: 967.6 / 967.1, 968.7 / 968.3, 550.8 / 551.0 tokens/s without / with -- the epilogue's chain behind the last block costs what the 6 us launch and
// its boundary cost): not shipped (EXPERIMENTS.md).
#pragma once
#include "../gemv_strip_cls.h"

namespace q4 {

// the greedy sampler's tail inside the classifier launch: argmax_kernel's arguments + the launch's own hand-off words
struct ClsArgmax {
    unsigned* counter;                 // arrival counter, zero between launches (the last arriver re-arms it)
    unsigned long long* cand;          // one {value bits, row} word per block
    int* result;                       // SharedData::tokens
    volatile int* pPos;                // SharedData::pos (pinned host word)
    int* pPosGpu;                      // RunState::pos
    int write_token;
    q4_half* x_next;                   // the next step's residual stream (the winner's embedding row goes there) or null
    const q4_half* table;
    int dim;
};
__device__ __forceinline__ void argmax_merge(float& v, int& ix, float ov, int op) {      // argmax_kernel's rule: value, then the lower index
    if (ov > v || (ov == v && op < ix)) { v = ov; ix = op; }
}

struct ClsArgmaxEpilogue {
    ClsArgmax am_;
    float best;          // (wave-uniform: wave_sum's result is)
    int best_row;
    int token_pos;
    __device__ __forceinline__ void entry(unsigned tid) {
        best = -INFINITY; best_row = 0x7fffffff; token_pos = 0;
        if (tid == 0) token_pos = *am_.pPosGpu;       // long landed when the epilogue wants it (the previous launch of the stream wrote it)
    }
    __device__ __forceinline__ void row(q4_half th, int r) {                   // ascending rows, strict: the first maximum stays (argmax_kernel :160-173)
        const float v = h2f(th);
        if (v > best) { best = v; best_row = r; }
    }
};

template <int NS, int D>
__global__ void __launch_bounds__(STRIP_WAVES * 64) cls_strip_argmax_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w, const unsigned wbytes,
                                                                           const unsigned rbase, const unsigned rrem, q4_half* __restrict__ out, const int n, const unsigned row_bytes,
                                                                           const ClsArgmax am) {
    using L = StripClsLds<NS, D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ClsArgmaxEpilogue epi;
    epi.am_ = am;
    cls_strip_body<NS, true, D>(arg_x, arg_rms, arg_w, wbytes, rbase, rrem, out, n, row_bytes, epi);
    const ClsArgmax& am_ = epi.am_;
    const float best = epi.best;
    const int best_row = epi.best_row;
    int token_pos = epi.token_pos;
    // ---- argmax_kernel as the epilogue of the launch
    float* sval = reinterpret_cast<float*>(smem + L::PART);                    // (the rmsnorm partials are long read)
    int* sidx = reinterpret_cast<int*>(smem + L::PART + 64);
    int* sflag = reinterpret_cast<int*>(smem + L::PART + 128);
    if (lane == 0) { sval[wave] = best; sidx[wave] = best_row; }
    __syncthreads();
    if (tid == 0) {
        float v = sval[0];
        int ix = sidx[0];
        for (int w = 1; w < STRIP_WAVES; w++) argmax_merge(v, ix, sval[w], sidx[w]);
        const unsigned long long c = (unsigned long long)(unsigned)as_i(v) | ((unsigned long long)(unsigned)ix << 32);
        unsigned long long* dst = am_.cand + blockIdx.x;
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(c) : "memory");   // written through, acknowledged
        const unsigned old = __hip_atomic_fetch_add(am_.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == gridDim.x - 1u;
        if (last) __hip_atomic_store(am_.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                      // re-armed for the next launch
        *sflag = last ? 1 : 0;
    }
    __syncthreads();
    if (*sflag == 0) return;
    // the block that arrived last: every candidate is in memory
    float v = -INFINITY;
    int ix = 0x7fffffff;
    for (unsigned b = tid; b < gridDim.x; b += STRIP_WAVES * 64) {
        const unsigned long long c = __hip_atomic_load(am_.cand + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        argmax_merge(v, ix, as_f((int)(unsigned)c), (int)(unsigned)(c >> 32));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int op = __shfl_xor(ix, off);
        argmax_merge(v, ix, ov, op);
    }
    __syncthreads();                                                           // (thread 0 has read the block's own totals)
    if (lane == 0) { sval[wave] = v; sidx[wave] = ix; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < STRIP_WAVES; w++) argmax_merge(v, ix, sval[w], sidx[w]);
        if (ix == 0x7fffffff) ix = 0;                                          // all NaN / -inf
        *sflag = ix;
    }
    if (am_.x_next != nullptr) {                                                // (uniform: a kernel argument) every other block is done with x
        __syncthreads();
        const int token = *sflag;
        // the upper waves copy the row while wave 0 publishes the token: two memory round trips side by side instead of in series
        for (int u = (int)tid - 512; u >= 0 && u < (am_.dim >> 3); u += 512)
            reinterpret_cast<u32x4*>(am_.x_next)[u] = reinterpret_cast<const u32x4*>(am_.table + (size_t)token * am_.dim)[u];
    }
    if (tid == 0) {
        token_pos++;
        if (am_.write_token) am_.result[token_pos] = ix;                         // gpu_kernels.h:486-487
        __threadfence_system();                                                // the host may be spinning on *pPos (q4_wait_pos)
        *am_.pPos = token_pos;                                                  // :490 (unblocks the CPU)
        *am_.pPosGpu = token_pos;                                               // :491
    }
}

}  // namespace q4
