// gemv_strip_cls.h -- the fp16 classifier GEMV (mat_vec_kernel, gpu_kernels.h:109-139; llama2_q4.cu:339) as strips, optionally with the final rmsnorm
// (rmsnorm_kernel, gpu_kernels.h:72-105; llama2_q4.cu:336) fused into its x staging. 262 MB at 7B: the one launch of a token that is pure streaming.
// One 16-wave block per CU owns a contiguous range of vocabulary rows; wave w takes rows w, w + 16, ... of it, streams each row (n halves = NS pieces
// of 1 KiB) with `buffer_load_dwordx4 ... nt lds` into its private four-piece ring (gemv_strip.h), and multiplies with x held in 4 NS registers. The
// arithmetic is gemv_f16_kernel's (q4_kernels.hip): per lane and row, for slots s = 0 .. NS - 1: acc = four v_dot2c from zero, sum += acc; wave_sum; one
// rounding -- so the public matmul and both forms here give the same bits. With NORM the block computes the norm once (the canonical reduction of
// q4_device.h, rms_apply8: the bits rmsnorm_kernel writes) instead of a launch of its own in front: 16,000 waves redoing it inside gemv_f16_kernel cost
// more than that launch (EXPERIMENTS.md notebook §9), 256 blocks do not. Measured (7B, one call): the launch with the norm inside 42.7 / 42.1 / 40.7 us at ring
// depth 2 / 4 / 8 against 40.2 + 4.7 us for gemv_f16_kernel behind rmsnorm_kernel -- a 262 MB stream sustains 6.2-6.5 TB/s on this chip whatever is in
// flight --, 969.3 -> 973.0 / 971.8 tokens/s at depth 4 / 8: depth 4 ships.
// An epilogue policy sees every logit a wave stores (exp/cls_argmax.h: the greedy sampler as this launch's epilogue, measured level and not shipped;
// the product's policy is empty).
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

struct ClsNoEpilogue {     // the product: nothing rides on the classifier launch
    __device__ __forceinline__ void entry(unsigned) {}
    __device__ __forceinline__ void row(q4_half, int) {}
};
template <int NS, bool NORM, int D, typename EPI>
__device__ __forceinline__ void cls_strip_body(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w, const unsigned wbytes,
                                               const unsigned rbase, const unsigned rrem, q4_half* __restrict__ out, const int n, const unsigned row_bytes, EPI& epi) {
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

    epi.entry(tid);
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
        epi.row(th, (int)(r0 + (unsigned)wave + 16u * (unsigned)i));           // (wave-uniform: wave_sum's result is)
    }
}
template <int NS, bool NORM, int D>
__global__ void __launch_bounds__(STRIP_WAVES * 64) cls_strip_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w, const unsigned wbytes,
                                                                    const unsigned rbase, const unsigned rrem, q4_half* __restrict__ out, const int n, const unsigned row_bytes) {
    ClsNoEpilogue none;
    cls_strip_body<NS, NORM, D>(arg_x, arg_rms, arg_w, wbytes, rbase, rrem, out, n, row_bytes, none);
}

// n = 4096 or 5120 inputs (8 or 10 pieces per row), contiguous rows, one right-hand side, alpha = 1, at least 64 rows per CU, a stream that may use every CU
static bool cls_strip_covers(int n, int d, int batch, int w_row_stride, float alpha) {
    if (g_gemv_form != GEMV_PRODUCT && g_gemv_form != GEMV_STRIPS_EVERYWHERE && g_gemv_form != GEMV_K5120_COLUMN_UNITS) return false;
    const int nb = cu_count();
    return (n == 4096 || n == 5120) && batch == 1 && w_row_stride == n && alpha == 1.0f && d / nb >= 64 && (long long)d * n * 2 < (1ll << 31) &&
           g_ablate == 0 && stream_cu_count() == nb;
}
constexpr int CLS_D = 4;   // ring depth: 74-76 KiB of LDS, opted in by cls_strip_prepare()
// (cls_strip_prepare, q4_kernels.hip: the LDS opt-in is not a stream operation -- build_transformer makes it, outside any capture)
template <int NS, bool NORM, int D>
static int launch_cls_strip_d(q4_half* out, const q4_half* x, const q4_half* rms_w, const q4_half* w, int n, int d) {
    { const int rc = cls_strip_prepare(); if (rc) return rc; }
    const unsigned nb = (unsigned)cu_count();
    constexpr size_t smem = StripClsLds<NS, D>::BYTES;
    Q4_LAUNCH((cls_strip_kernel<NS, NORM, D>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(rms_w),
              (const void*)w, (unsigned)((size_t)d * n * 2), (unsigned)d / nb, (unsigned)d % nb, out, n, (unsigned)n * 2u);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
// rms_w != nullptr: out = W . rmsnorm(x, rms_w) (x itself is left as it is); else out = W . x
static int launch_cls_strip(q4_half* out, const q4_half* x, const q4_half* rms_w, const q4_half* w, int n, int d) {
    if (n == 4096) return rms_w ? launch_cls_strip_d<8, true, CLS_D>(out, x, rms_w, w, n, d) : launch_cls_strip_d<8, false, CLS_D>(out, x, nullptr, w, n, d);
    return rms_w ? launch_cls_strip_d<10, true, CLS_D>(out, x, rms_w, w, n, d) : launch_cls_strip_d<10, false, CLS_D>(out, x, nullptr, w, n, d);
}

}  // namespace q4
