// exp/qkv_strip.h -- LABORATORY (libllama2_q4_prof.so only; knob 11 = 8): bit-identical, measured slower than the wave-owned q/k/v kernel (EXPERIMENTS.md).
// gemv_strip_qkv.h -- the fused q/k/v launch (rmsnorm_kernel + qkv_matvec_kernel + RoPERotation_kernel, gpu_kernels.h:72-105, 242-254, 332-355;
// llama2_q4.cu:300, 307, 317) as strips, for the shape where gemv_q4.h's grid leaves the CUs uneven: Llama-2-13B (K = N = 5120: 960 four-wave blocks
// = 3.75 per CU, the launch pays for four rows of blocks). Same scheme as gemv_strip.h -- one 16-wave block per CU, every wave streams its own units
// with `buffer_load_dwordx4 ... nt lds` into a private ring of two 1 KiB pieces, waits with vmcnt, reads back, re-issues, multiplies -- with these
// differences:
//   * a CU owns TEN RoPE pairs (i, i + head_size / 2) of each of the three matrices = 60 (column, matrix) units of 2.5 KiB; waves 0-4 take q, 5-9 k,
//     10-14 v, wave 5 m + j the pairs j and j + 5 of the CU: four units per wave, in gemv_q4.h's row order (pair A first, pair B first, A second, B
//     second), so the transposing reduction and the RoPE epilogue are that kernel's own, inside the wave; wave 15 only helps to stage x;
//   * a pair's columns are head_size / 2 apart and a CU's ten pairs may straddle two heads: scales and zeros arrive as dword-granular LDS-DMA
//     gathers (`buffer_load_dword ... lds`: 64 arbitrary dwords per instruction), 27 instructions per CU;
//   * the position is requested at entry, the (cos, sin) entries of a wave's rows as soon as it is known (one more load between the ring's first
//     pieces: the first two waits count it), so nothing dependent is left for the epilogue;
//   * a column's half piece goes to the lower half of the wave for even pairs and to the upper half for odd ones: the lanes and operations of
//     gemv_q4.h's shared half slot (there: pair 2 wg below, pair 2 wg + 1 above).
// Same arithmetic in the same order as gemv_q4_kernel<MODE_QKV, 3, 4, NORM, 0, 1, true>: bit for bit (tests/prof_cases.py) -- and SLOWER: 9.65 against
// 8.97 us per launch by HIP events inside the eager 13B network (tools/lab/qkv_strip_time.py), 565.1 -> 558.0 and 561.0 -> 551.8 tokens/s at 13B -n 256
// (knob 11 = 16 against 0 while the form was the product's choice, one process each). Strips stream 160 KB per CU at the rate the 13B down
// projection's strips reach (144 KB in 8.4 us); that beat a K-split grid of 2.5 blocks per CU (10.5 us) and does not beat this launch's 3.75 blocks
// per CU. Profiling build only, knob 11 = 8 (EXPERIMENTS.md notebook §9.20).
#pragma once
#include "../gemv_strip.h"

namespace q4 {

constexpr int SQ_PAIRS = 10, SQ_COLS = 2 * SQ_PAIRS;       // per CU and matrix
struct StripQkvLds {
    static constexpr unsigned RING = 0;                                 // [16 waves][2] x 1 KiB
    static constexpr unsigned SIDE_S_BYTES = 7u * 256u;                 // per matrix: 20 columns x 40 groups x 2 B = 1600
    static constexpr unsigned SIDE_S = RING + STRIP_WAVES * 2048u;
    static constexpr unsigned SIDE_Z_BYTES = 2u * 256u;                 // per matrix: 20 columns x 5 words x 4 B = 400
    static constexpr unsigned SIDE_Z = SIDE_S + 3 * SIDE_S_BYTES;
    static constexpr unsigned XS = SIDE_Z + 3 * SIDE_Z_BYTES;           // [3][4][64] x 16 B permuted x
    static constexpr unsigned SX = XS + 3 * 4096u;                      // [3][64] -(sum of the 32 x) * 2^-20
    static constexpr unsigned PART = SX + 3 * 256u;                     // [768] rmsnorm chunk partials (zero past K / 8)
    static constexpr unsigned BYTES = PART + 3 * 1024u;
};

// 64 dwords from (descriptor, voff per lane) to LDS bytes [lds_dst, lds_dst + 256): lane l lands at lds_dst + 4 l
__device__ __forceinline__ void dma_dwords_default(unsigned lds_dst, unsigned voff, __amdgpu_buffer_rsrc_t r) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_dst), "v"(voff), "s"(r) : "memory");
}

template <bool NORM>
__global__ void __launch_bounds__(STRIP_WAVES * 64) qkv_strip_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const int* __restrict__ arg_pos,
                                                                     const unsigned wbytes, const GemvArgs a) {
    constexpr unsigned CB = 2560u, G = 40u, ZW = 5u;   // bytes, quantisation groups and words of zero nibbles of a column (K = 5120)
    constexpr int NSTAGE = 10, TS = 3, D = 2;
    using L = StripQkvLds;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stager = wave < NSTAGE, worker = wave < 15;
    const int mat = wave / 5, jw = wave - 5 * mat;      // this wave's matrix (3: none) and its first pair of the CU's ten
    const unsigned hp = (unsigned)a.head_size >> 1;
    const unsigned p0 = blockIdx.x * (unsigned)SQ_PAIRS;
    const unsigned head0 = p0 / hp, i0 = p0 - head0 * hp;       // (uniform) the CU's ten pairs lie in head0 and, past its last pair, in head0 + 1
    auto pair_index = [&](unsigned q) -> unsigned { const unsigned i = i0 + q; return i >= hp ? i - hp : i; };   // q = 0 .. 9: index inside its head
    // local column lc of the CU: pair lc % 10, second column of the pair for lc >= 10
    auto column_of = [&](unsigned lc) -> unsigned {
        const bool second = lc >= (unsigned)SQ_PAIRS;
        const unsigned i = i0 + (second ? lc - (unsigned)SQ_PAIRS : lc);
        const bool wrap = i >= hp;
        return (head0 + (wrap ? 1u : 0u)) * (unsigned)a.head_size + (wrap ? i - hp : i) + (second ? hp : 0u);
    };
    const unsigned voff = lane * 16u;
    const bool upper = lane >= 32u;

    // ---- what this wave will wait for first: the position, x, the side data, then its ring
    int posv = 0;
    asm volatile("global_load_dword %0, %1, off" : "=&v"(posv) : "v"(arg_pos) : "memory");
    u32x4 xraw = {0u, 0u, 0u, 0u}, wraw = {0u, 0u, 0u, 0u};
    if (stager) {                                      // asm loads: hipcc must not count them (it cannot see the DMA pieces behind them)
        const u32x4* px = arg_x + tid;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xraw) : "v"(px) : "memory");
        if (NORM) {
            const u32x4* pw = arg_rms + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wraw) : "v"(pw) : "memory");
        }
    }
    // scales and zeros of the CU's 3 x 20 columns: 27 gathers of 64 dwords (per matrix 7 of scales, 2 of zeros), instruction t by wave t % 16
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const int t = wave + 16 * rep;
        if (t < 27) {
            const int m = t / 9, k = t - 9 * m;
            if (k < 7) {
                unsigned e = (unsigned)k * 64u + lane;
                e = e < SQ_COLS * (G / 2u) ? e : SQ_COLS * (G / 2u) - 1u;          // 400 dwords
                const unsigned lc = e / (G / 2u), wd = e - lc * (G / 2u);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].s, 0, a.N * a.sh * 2, 0x00020000);
                dma_dwords_default(L::SIDE_S + (unsigned)m * L::SIDE_S_BYTES + (unsigned)k * 256u, column_of(lc) * (G * 2u) + wd * 4u, rs);
            } else {
                unsigned e = (unsigned)(k - 7) * 64u + lane;
                e = e < SQ_COLS * ZW ? e : SQ_COLS * ZW - 1u;                        // 100 dwords
                const unsigned lc = e / ZW, wd = e - lc * ZW;
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].z, 0, a.N * a.pzh * 4, 0x00020000);
                dma_dwords_default(L::SIDE_Z + (unsigned)m * L::SIDE_Z_BYTES + (unsigned)(k - 7) * 256u, column_of(lc) * (ZW * 4u) + wd * 4u, rz);
            }
        }
    }
    block_barrier_lds();      // the x loads are queued on this CU in front of every weight piece (the path returns in order)

    // this wave's units, rows of the reduction: r = 0 pair A first, 1 pair B first, 2 A second, 3 B second (A = pair jw, B = pair jw + 5 of the CU)
    const int mm = worker ? mat : 0;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.m[mm].w), 0, (int)wbytes, 0x00020000);
    unsigned lcr[4], soff[4];
    bool odd[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        lcr[r] = (unsigned)(jw + ((r & 1) ? 5 : 0) + ((r & 2) ? SQ_PAIRS : 0));
        soff[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)(column_of(lcr[r]) * CB));
        odd[r] = (((p0 + (unsigned)jw + ((r & 1) ? 5u : 0u)) & 1u) != 0u);
    }
    const unsigned ring = L::RING + (unsigned)wave * (D * 1024u);
    const unsigned voff_half = (lane & 31u) * 16u;
    auto issue2 = [&](int r, int ks) {                  // piece ks of unit r (constants at every call site)
        const unsigned dst = ring + (unsigned)((TS * r + ks) & (D - 1)) * 1024u, so = soff[r] + (unsigned)ks * 1024u;
        if (ks == 2) { if (upper == odd[r]) dma_piece(dst, voff_half, rw, so); }   // 32 lanes: the LDS address follows the LANE, not the offset
        else dma_piece(dst, voff, rw, so);
    };
    if (worker) { issue2(0, 0); issue2(0, 1); }

    // ---- x chain (gemv_q4_body's staging, one 8-half chunk per thread of waves 0..9)
    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    if (worker) asm volatile("s_waitcnt vmcnt(2)" : "+v"(xraw), "+v"(wraw), "+v"(posv) : : "memory");   // all but the two weight pieces
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xraw), "+v"(wraw), "+v"(posv) : : "memory");
    const int pos = __builtin_amdgcn_readfirstlane(posv);
    // (cos, sin) of this lane's row: RoPERotation_kernel's angle for pair index i at this position, from the model's table
    const int row = (int)(lane >> 4);
    const unsigned irow = pair_index((unsigned)jw + ((row & 1) ? 5u : 0u));
    u32x2v csraw = {0u, 0u};
    const bool roped = mat < 2;                        // (a.rope holds: qkv_strip_covers; the v waves load their entries too -- one wait schedule for all)
    if (worker) {
        const float2* pt = a.rope_table + (size_t)pos * hp + irow;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(csraw) : "v"(pt) : "memory");
    }
    if (NORM) {
        if (tid < (unsigned)(TS * 256)) part[tid] = stager ? sumsq8(xraw, 0.f) : 0.f;      // (entries 640 .. 767 are the zero padding of the canonical sum)
        block_barrier_lds();
    }
    if (stager) {
        float ss = 1.f;
        if (NORM) ss = rms_scale_from_partials<TS * 256>(part, TS * 256, a.K);
        u32x4 v = xraw;
        const unsigned sgn = q4_stage_sign_bits(tid);      // odd units are staged negated (gemv_q4.h, q4_stage_sign_bits)
        if (NORM) v = rms_apply8(v, wraw, q4_signed_scale(ss, sgn));
        else v = q4_signed_x(v, sgn);
        const u32x4 pv = permute_x8(v);
        const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
        float cb = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
        cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);   // quad sum: the 32 inputs of one uint4 unit
        const unsigned j = tid >> 2, d = tid & 3u;
        xs[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
        if (d == 0) sx[j] = cb * -9.5367431640625e-07f;     // -(sum x) * 2^-20
    }
    block_barrier_lds();                               // x staged; side data landed (its issuers passed the vmcnt wait above)
    if (!worker) return;
    u32x4 X[TS][4];
    float corr[TS];
#pragma unroll
    for (int ks = 0; ks < TS; ks++) {
        const unsigned lu = ks == 2 ? (lane & 31u) : lane;                   // half slot: both halves of the wave hold units 0-31
#pragma unroll
        for (int d = 0; d < 4; d++) X[ks][d] = xs[((ks * 4 + d) << 6) + lu];
        corr[ks] = sx[ks * 64 + lu];
    }
    const unsigned char* wbase = smem + ring + lane * 16u;
    const unsigned char* sbase = smem + L::SIDE_S + (unsigned)mat * L::SIDE_S_BYTES + (lane >> 2) * 2u;     // + lc * 80
    const unsigned char* zbase = smem + L::SIDE_Z + (unsigned)mat * L::SIDE_Z_BYTES + (lane >> 5) * 4u;     // + lc * 20
    const unsigned zsh = ((lane >> 2) & 7u) * 4u;

    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float c = 0.f;
#pragma unroll
        for (int ks = 0; ks < TS; ks++) {
            const int j = TS * r + ks;                  // 0 .. 11
            const int e = j & (D - 1);
            const bool hs = ks == 2;                    // the half slot
            // in issue order behind piece j: piece j + 1 and, while j < 2, the (cos, sin) load that went out between pieces 1 and 2
            if (j < 2) wait_vmcnt<2>();
            else if (j + D < TS * 4) wait_vmcnt<1>();
            else wait_vmcnt<0>();
            const u32x4 w = *reinterpret_cast<const u32x4*>(wbase + e * 1024);
            // scale and zero word of this lane's group: 16 ks + lane / 4, in the half slot 32 + (lane % 32) / 4
            const uint16_t sc = *reinterpret_cast<const uint16_t*>(sbase + lcr[r] * (G * 2u) + (hs ? 64 - (int)(upper ? 16u : 0u) : ks * 32));
            const unsigned zw = *reinterpret_cast<const unsigned*>(zbase + lcr[r] * (ZW * 4u) + (hs ? 16 - (int)(upper ? 4u : 0u) : ks * 8));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done: the entry may be refilled
            if (j + D < TS * 4) issue2((j + D) / TS, (j + D) % TS);
            float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const unsigned ww = w[d];
                const unsigned tt = ww >> 8;
                acc_e = __builtin_amdgcn_fdot2(as_h2(ww & 0x000F000Fu), as_h2(X[ks][d][0]), acc_e, false);
                acc_o = __builtin_amdgcn_fdot2(as_h2(ww & 0x00F000F0u), as_h2(X[ks][d][1]), acc_o, false);
                acc_e = __builtin_amdgcn_fdot2(as_h2(tt & 0x000F000Fu), as_h2(X[ks][d][2]), acc_e, false);
                acc_o = __builtin_amdgcn_fdot2(as_h2(tt & 0x00F000F0u), as_h2(X[ks][d][3]), acc_o, false);
            }
            const float zf = (float)((zw >> zsh) & 0xFu);
            float t = __builtin_fmaf(acc_e, 16.f, acc_o);
            t = __builtin_fmaf(zf, corr[ks], t);
            if (hs) {       // gemv_q4.h's half slot: a product and a sum (not an fma), only on the half of the wave that serves this column
                const float v = h2f(sc) * t;
                c += (upper == odd[r]) ? v : 0.f;
            } else {
                c = __builtin_fmaf(h2f(sc), t, c);
            }
            asm volatile("" : "+v"(c));                 // a piece's arithmetic stays in front of the next piece's wait (left alone hipcc sinks the arithmetic of all twelve
                                                        // pieces behind the last wait and parks the pieces in registers: 64 bytes of scratch)
        }
        cs[r] = c;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(csraw) : : "memory");   // (long landed: the third wait above covered it)

    // ---- gemv_q4.h's MODE_QKV epilogue: row r of the wave holds unit r's total; RoPE on the fp16-rounded outputs of q and k
    q4_half* out = a.out[mat];
    if (mat != 0) out += (size_t)a.loff + (size_t)pos * a.N;                          // gpu_kernels.h:251,253
    const float mine = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;
    float r = mine;
    if (roped) {
        const float other = round_h(__shfl_xor(mine, 32));
        const float me = round_h(mine);
        const float fcr = as_f((int)csraw[0]), fci = as_f((int)csraw[1]);
        r = (row & 2) ? (other * fci + me * fcr) : (me * fcr - other * fci);           // :345-346
    }
    if ((lane & 15u) == 0) {
        const unsigned lc = (unsigned)jw + ((row & 1) ? 5u : 0u) + ((row & 2) ? (unsigned)SQ_PAIRS : 0u);
        out[column_of(lc)] = f2h(r);
    }
    // the next launch of the stream (attention -> o-proj, layer_attn.h) tags its hand-off granules with this word (gemv_q4.h, `bump`)
    if (a.bump != nullptr && blockIdx.x == 0 && tid == 0) __hip_atomic_fetch_add(a.bump, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Llama-2-13B's shape: K = N = 5120 with the shared half slot, multi-head (N_kv = N), ten pairs per CU, the RoPE table of the model, on a stream
// that may use every CU. Not the product's choice (measured slower, see the header): g_gemv_form = GEMV_STRIPS_EVERYWHERE only (profiling build: strips wherever covered).
static bool qkv_strip_covers(const GemvArgs& a) {
    if (g_gemv_form != GEMV_STRIPS_EVERYWHERE) return false;
    const int nb = cu_count();
    return strip_k5120(a) && a.nslots == 3 && half_tail(a) && a.N_kv == 0 && a.N == 2 * SQ_PAIRS * nb && a.rope && a.rope_table != nullptr && a.head_size >= 4 &&
           (a.head_size & 3) == 0 && a.N % a.head_size == 0 && a.pPos != nullptr && g_ablate == 0 && stream_cu_count() == nb;
}
static int launch_qkv_strip(const GemvArgs& a) {
    constexpr size_t smem = StripQkvLds::BYTES;
    static_assert(smem <= 64 * 1024, "no opt-in: capturable as it is");
    const unsigned nb = (unsigned)cu_count();
    if (a.rms_w)
        Q4_LAUNCH((qkv_strip_kernel<true>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w), a.pPos,
                  (unsigned)(a.N * a.pw4 * 16), a);
    else
        Q4_LAUNCH((qkv_strip_kernel<false>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w), a.pPos,
                  (unsigned)(a.N * a.pw4 * 16), a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // namespace q4
