// gemv_strip_cls.h -- the fp16 classifier GEMV (mat_vec_kernel, gpu_kernels.h:109-139; llama2_q4.cu:339) as strips, optionally with the final rmsnorm
// (rmsnorm_kernel, gpu_kernels.h:72-105; llama2_q4.cu:336) fused into its x staging. 262 MB at 7B: the one launch of a token that is pure streaming.
// One 16-wave block per CU owns a contiguous range of vocabulary rows; wave w takes rows w, w + 16, ... of it, streams each row (n halves = NS pieces
// of 1 KiB) with `buffer_load_dwordx4 ... nt lds` into its private four-piece ring (gemv_strip.h), and multiplies with x held in 4 NS registers. The
// arithmetic is gemv_f16_kernel's (q4_kernels.hip): per lane and row, for slots s = 0 .. NS - 1: acc = four v_dot2c from zero, sum += acc; wave_sum; one
// rounding -- so the public matmul and both forms here give the same bits. With NORM the block computes the norm once (the canonical reduction of
// q4_device.h, rms_apply8: the bits rmsnorm_kernel writes) instead of a launch of its own in front: 16,000 waves redoing it inside gemv_f16_kernel cost
// more than that launch, 256 blocks do not. Measured (7B, one call): the launch with the norm inside 42.7 / 42.1 / 40.7 us at ring
// depth 2 / 4 / 8 against 40.2 + 4.7 us for gemv_f16_kernel behind rmsnorm_kernel -- a 262 MB stream sustains 6.2-6.5 TB/s on this chip whatever is in
// flight --, 969.3 -> 973.0 / 971.8 tokens/s at depth 4 / 8: depth 4 ships.
// AM: the greedy sampler (argmax_kernel, gpu_kernels.h:448-493; llama2_q4.cu:384 -> sampler.h:47-49) as this launch's epilogue. Every wave keeps the
// best of its own rows (the fp16-rounded logit it stores, first maximum = lowest row), a block leaves ONE candidate {value, row} (written through,
// drained) and one returning arrival; the block that arrives last reads the 256 candidates, decides with argmax_kernel's rule (value, then the lower
// index), copies the winner's embedding row for the next step where asked, writes the token ring and advances both position words. One launch and one
// boundary fewer per greedy token; the logits are stored as before.
#pragma once
#include "gemv_strip.h"

namespace q4 {

template <int NS, int D>
struct StripClsLds {
    static constexpr unsigned RING = 0;                                 // [16 waves][D] x 1 KiB
    static constexpr unsigned XN = RING + STRIP_WAVES * D * 1024u;      // [NS][64] x 16 B: the (normalised) input, chunk j = s * 64 + lane
    static constexpr unsigned PART = XN + NS * 1024u;                   // [NS * 64] rmsnorm chunk partials
    static constexpr unsigned BYTES = PART + NS * 256u;
};

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

template <int NS, bool NORM, int D, bool AM = false>
__global__ void __launch_bounds__(STRIP_WAVES * 64) cls_strip_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w, const unsigned wbytes,
                                                                    const unsigned rbase, const unsigned rrem, q4_half* __restrict__ out, const int n, const unsigned row_bytes,
                                                                    const ClsArgmax am) {
    using L = StripClsLds<NS, D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned r0 = blockIdx.x * rbase + (blockIdx.x < rrem ? blockIdx.x : rrem);
    const int nr = (int)(rbase + (blockIdx.x < rrem ? 1u : 0u));
    const int nu = (nr - wave + STRIP_WAVES - 1) / STRIP_WAVES;          // this wave's rows: r0 + wave + 16 i
    const int npieces = NS * nu;
    const unsigned voff = lane * 16u;
    const bool stager = wave < NS;                                       // one 8-half chunk of x per thread of waves 0 .. NS - 1

    u32x4 xraw = {0u, 0u, 0u, 0u}, wraw = {0u, 0u, 0u, 0u};
    if (stager) {                                                        // asm loads: hipcc must not count them (it cannot see the DMA pieces behind them)
        const u32x4* px = arg_x + tid;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xraw) : "v"(px) : "memory");
        if (NORM) {
            const u32x4* pw = arg_rms + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wraw) : "v"(pw) : "memory");
        }
    }
    block_barrier_lds();      // the x loads are queued on this CU in front of every weight piece (the path returns in order)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(arg_w), 0, (int)wbytes, 0x00020000);
    const unsigned ring = L::RING + (unsigned)wave * (D * 1024u);
    const unsigned soff0 = (r0 + (unsigned)wave) * row_bytes;             // piece s of row i: soff0 + i * 16 rows + s * 1024
    auto issue2 = [&](int i, int s) { dma_piece(ring + (unsigned)((NS * i + s) & (D - 1)) * 1024u, voff, rw, soff0 + (unsigned)i * (16u * row_bytes) + (unsigned)s * 1024u); };
    static_assert(D == 2 || D == 4 || D == 8, "ring depth");
    static_assert(D <= NS, "the first D pieces are the first row's");
#pragma unroll
    for (int k = 0; k < D; k++)
        if (k < npieces) issue2(0, k);

    // ---- x chain: rmsnorm_kernel's arithmetic on this block's copy of x, or x as it is
    u32x4* xn = reinterpret_cast<u32x4*>(smem + L::XN);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    if (npieces >= D) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xraw), "+v"(wraw) : "n"(D) : "memory");   // all but the D weight pieces
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xraw), "+v"(wraw) : : "memory");
    if (NORM) {
        if (stager) part[tid] = sumsq8(xraw, 0.f);
        block_barrier_lds();
    }
    if (stager) {
        u32x4 v = xraw;
        if (NORM) v = rms_apply8(v, wraw, rms_scale_from_partials<NS * 64>(part, NS * 64, n));
        xn[tid] = v;
    }
    block_barrier_lds();
    u32x4 X[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) X[s] = xn[s * 64 + lane];
    const unsigned char* wbase = smem + ring + lane * 16u;

    float best = -INFINITY;          // (wave-uniform: wave_sum's result is)
    int best_row = 0x7fffffff;
    int token_pos = 0;
    if (AM && tid == 0) token_pos = *am.pPosGpu;       // long landed when the epilogue wants it (the previous launch of the stream wrote it)
    for (int i = 0; i < nu; i++) {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int j = NS * i + s;
            if (j + D < npieces) wait_vmcnt<D - 1>(); else wait_vmcnt<0>();    // piece j has landed
            const u32x4 w = *reinterpret_cast<const u32x4*>(wbase + ((NS * i + s) & (D - 1)) * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the read is done: the entry may be refilled
            if (j + D < npieces) issue2(i + (s + D) / NS, (s + D) % NS);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_fdot2(as_h2(w[e]), as_h2(X[s][e]), acc, false);
            sum += acc;
        }
        float t = wave_sum(sum);
        t *= 1.0f;                                                             // alpha, gpu_kernels.h:135
        const q4_half th = f2h(t);
        if (lane == 0) out[r0 + (unsigned)wave + 16u * (unsigned)i] = th;
        if (AM) {                                                              // ascending rows, strict: the first maximum stays (argmax_kernel :160-173)
            const float v = h2f(th);
            if (v > best) { best = v; best_row = (int)(r0 + (unsigned)wave + 16u * (unsigned)i); }
        }
    }
    if (!AM) return;
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
        unsigned long long* dst = am.cand + blockIdx.x;
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(c) : "memory");   // written through, acknowledged
        const unsigned old = __hip_atomic_fetch_add(am.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == gridDim.x - 1u;
        if (last) __hip_atomic_store(am.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                      // re-armed for the next launch
        *sflag = last ? 1 : 0;
    }
    __syncthreads();
    if (*sflag == 0) return;
    // the block that arrived last: every candidate is in memory
    float v = -INFINITY;
    int ix = 0x7fffffff;
    for (unsigned b = tid; b < gridDim.x; b += STRIP_WAVES * 64) {
        const unsigned long long c = __hip_atomic_load(am.cand + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    if (am.x_next != nullptr) {                                                // (uniform: a kernel argument) every other block is done with x
        __syncthreads();
        const int token = *sflag;
        // the upper waves copy the row while wave 0 publishes the token: two memory round trips side by side instead of in series
        for (int u = (int)tid - 512; u >= 0 && u < (am.dim >> 3); u += 512)
            reinterpret_cast<u32x4*>(am.x_next)[u] = reinterpret_cast<const u32x4*>(am.table + (size_t)token * am.dim)[u];
    }
    if (tid == 0) {
        token_pos++;
        if (am.write_token) am.result[token_pos] = ix;                         // gpu_kernels.h:486-487
        __threadfence_system();                                                // the host may be spinning on *pPos (q4_wait_pos)
        *am.pPos = token_pos;                                                  // :490 (unblocks the CPU)
        *am.pPosGpu = token_pos;                                               // :491
    }
}

// n = 4096 or 5120 inputs (8 or 10 pieces per row), contiguous rows, one right-hand side, alpha = 1, at least 64 rows per CU, a stream that may use every CU
static bool cls_strip_covers(int n, int d, int batch, int w_row_stride, float alpha) {
    if (g_engine != 0 && g_engine != 8 && g_engine != 19) return false;
    const int nb = cu_count();
    return (n == 4096 || n == 5120) && batch == 1 && w_row_stride == n && alpha == 1.0f && d / nb >= 64 && (long long)d * n * 2 < (1ll << 31) &&
           g_ablate == 0 && stream_cu_count() == nb;
}
constexpr int CLS_D = 4;   // ring depth: 74-76 KiB of LDS, opted in by cls_strip_prepare()
// the LDS opt-in is not a stream operation: q4_set_device and build_transformer make it (q4_runtime.hip), outside any capture
int cls_strip_prepare() {
    static bool opted = false;
    if (!opted) {
        Q4_HIP(hipFuncSetAttribute((const void*)cls_strip_kernel<8, true, CLS_D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StripClsLds<8, CLS_D>::BYTES));
        Q4_HIP(hipFuncSetAttribute((const void*)cls_strip_kernel<8, false, CLS_D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StripClsLds<8, CLS_D>::BYTES));
        Q4_HIP(hipFuncSetAttribute((const void*)cls_strip_kernel<10, true, CLS_D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StripClsLds<10, CLS_D>::BYTES));
        Q4_HIP(hipFuncSetAttribute((const void*)cls_strip_kernel<10, false, CLS_D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StripClsLds<10, CLS_D>::BYTES));
#ifdef Q4_PROFILING     // the sampler-epilogue form exists in the profiling build only (knob 12)
        Q4_HIP(hipFuncSetAttribute((const void*)cls_strip_kernel<8, true, CLS_D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StripClsLds<8, CLS_D>::BYTES));
        Q4_HIP(hipFuncSetAttribute((const void*)cls_strip_kernel<10, true, CLS_D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StripClsLds<10, CLS_D>::BYTES));
#endif
        opted = true;
    }
    return Q4_OK;
}
template <int NS, bool NORM, int D, bool AM>
static int launch_cls_strip_d(q4_half* out, const q4_half* x, const q4_half* rms_w, const q4_half* w, int n, int d, const ClsArgmax& am) {
    { const int rc = cls_strip_prepare(); if (rc) return rc; }
    const unsigned nb = (unsigned)cu_count();
    constexpr size_t smem = StripClsLds<NS, D>::BYTES;
    Q4_LAUNCH((cls_strip_kernel<NS, NORM, D, AM>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(rms_w),
              (const void*)w, (unsigned)((size_t)d * n * 2), (unsigned)d / nb, (unsigned)d % nb, out, n, (unsigned)n * 2u, am);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
// rms_w != nullptr: out = W . rmsnorm(x, rms_w) (x itself is left as it is); else out = W . x. am != nullptr (with rms_w): the greedy sampler as the
// launch's epilogue
static int launch_cls_strip(q4_half* out, const q4_half* x, const q4_half* rms_w, const q4_half* w, int n, int d, const ClsArgmax* am) {
    const ClsArgmax none = {};
#ifdef Q4_PROFILING
    if (am && rms_w) return n == 4096 ? launch_cls_strip_d<8, true, CLS_D, true>(out, x, rms_w, w, n, d, *am) : launch_cls_strip_d<10, true, CLS_D, true>(out, x, rms_w, w, n, d, *am);
#else
    (void)am;
#endif
    if (n == 4096) return rms_w ? launch_cls_strip_d<8, true, CLS_D, false>(out, x, rms_w, w, n, d, none) : launch_cls_strip_d<8, false, CLS_D, false>(out, x, nullptr, w, n, d, none);
    return rms_w ? launch_cls_strip_d<10, true, CLS_D, false>(out, x, rms_w, w, n, d, none) : launch_cls_strip_d<10, false, CLS_D, false>(out, x, nullptr, w, n, d, none);
}

}  // namespace q4
