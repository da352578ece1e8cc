// gemv_strip.h -- "strips": the fused gate/up GEMV (rmsnorm_kernel + ffn_matvec_silu_kernel, gpu_kernels.h:72-105, 256-275) on LDS-DMA rings. Ships
// for wide matrices (ffn_strip_covers below). NO loader wave: one 16-wave block per CU owns a contiguous range of columns;
// every wave streams its OWN units -- unit u = wave + 16 i of the block's (column, matrix) pairs, 2 KiB each at K = 4096, 2.5 KiB at K = 5120 -- with
// `buffer_load_dwordx4 ... nt lds` into a private ring of two 1 KiB pieces (lds_dma.h), waits for its oldest piece with vmcnt (a wave's loads return
// in order), reads it back with ds_read_b128, re-issues and multiplies with the denormal-nibble v_dot2c body of gemv_q4.h. Four waves per SIMD, x staged
// once per CU by waves 0..7 (0..9) and held in 32 (48) registers by every wave. Same arithmetic in the same order as gemv_q4_kernel<MODE_FFN>, bit for
// bit (tests/prof_cases.py). Deeper rings, the paced issue and the stamped build live in exp/ffn_strip_variants.h (profiling library only).
#pragma once
#include "gemv_q4.h"
#include "lds_dma.h"

namespace q4 {

constexpr int STRIP_WAVES = 16, STRIP_NCMAX = 56;
// TS = k-slots of a column: 2 (K = 4096: two 1 KiB pieces) or 3 (K = 5120: two pieces and a half one, the shared half slot of gemv_q4.h)
template <int D, int TS = 2>
struct StripLds {
    static constexpr unsigned NS_S = TS == 2 ? 4u : 5u, NS_Z = TS == 2 ? 1u : 2u;   // 1 KiB side pieces per matrix
    static constexpr unsigned RING = 0;                                      // [16 waves][D] x 1 KiB
    static constexpr unsigned SIDE_S_BYTES = NS_S * 1024u;                   // per matrix: 56 columns x 32 (40) groups x 2 B = 3584 (4480)
    static constexpr unsigned SIDE_S = RING + STRIP_WAVES * D * 1024u;
    static constexpr unsigned SIDE_Z_BYTES = NS_Z * 1024u;                   // per matrix: 56 columns x 4 (5) words x 4 B = 896 (1120)
    static constexpr unsigned SIDE_Z = SIDE_S + 2 * SIDE_S_BYTES;
    static constexpr unsigned XS = SIDE_Z + 2 * SIDE_Z_BYTES;                // [TS][4][64] x 16 B permuted x
    static constexpr unsigned SX = XS + TS * 4096u;                          // [TS][64] -(sum of the 32 x) * 2^-20
    static constexpr unsigned PART = SX + TS * 256u;                         // [TS * 256] rmsnorm chunk partials (zero past K / 8)
    static constexpr unsigned TOT = PART + TS * 1024u;                       // [NCMAX][2] column totals
    static constexpr unsigned STAMP = TOT + 512u;                            // (laboratory variants, exp/ffn_strip_variants.h: [64] wall-clock stamps,
    static constexpr unsigned FLAGS = STAMP + 512u;                          //  the paced forms' flag words; the product leaves both untouched)
    static constexpr unsigned BYTES = FLAGS + 64u;
};

template <bool NORM, int TS = 2>
__global__ void __launch_bounds__(STRIP_WAVES * 64) ffn_strip_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w0, const void* arg_w1, const unsigned wbytes,
                                                                     const unsigned cbase, const unsigned crem, const GemvArgs a) {
    // the scalars the entry needs come first: built with -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave instead of
    // through a scalar load from the kernel-argument segment
    constexpr int D = 2;                               // ring depth: two 1 KiB pieces per wave, 32 KiB in flight per CU (deeper rings measured level or slower)
    constexpr unsigned CB = TS == 2 ? 2048u : 2560u;   // bytes of a column
    constexpr unsigned G = TS == 2 ? 32u : 40u, ZW = TS == 2 ? 4u : 5u;   // its quantisation groups (one fp16 scale each), its words of zero nibbles
    constexpr int NSTAGE = TS == 2 ? 8 : 10;           // waves that stage x: one 8-half chunk per thread
    using L = StripLds<D, TS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned c0 = blockIdx.x * cbase + (blockIdx.x < crem ? blockIdx.x : crem);
    const int nc = (int)(cbase + (blockIdx.x < crem ? 1u : 0u));
    const int mat = wave & 1;
    const int nu = (2 * nc - wave + 15) >> 4;          // this wave's units: u = wave + 16 i, column c0 + u / 2, matrix u % 2
    const int npieces = TS * nu;
    const unsigned voff = lane * 16u;
    const bool stager = wave < NSTAGE;
    // K = 5120: a column's third piece is 512 bytes, 32 lanes. As in gemv_q4.h's shared half slot the lower half of the wave takes it
    // for the first column of a pair (even), the upper half for the second (odd): the same lanes add the same terms
    const bool upper = lane >= 32u;

    // ---- what this wave will wait for first: x (its address arrives in SGPRs when the build preloads kernel arguments), side data, then its ring
    u32x4 xraw = {0u, 0u, 0u, 0u}, wraw = {0u, 0u, 0u, 0u};
    if (stager) {                                      // asm loads: hipcc must not count them (it cannot see the DMA pieces behind them)
        const u32x4* px = arg_x + tid;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xraw) : "v"(px) : "memory");
        if (NORM) {
            const u32x4* pw = arg_rms + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wraw) : "v"(pw) : "memory");
        }
    }
    constexpr int NSIDE = (int)(L::NS_S + L::NS_Z);    // side pieces per matrix: scales (4 or 5 KiB) and zeros (1 or 2 KiB) of the block's columns
    if (wave < 2 * NSIDE) {
        const int m = wave >= NSIDE, p = wave - NSIDE * m;
        if (p < (int)L::NS_S) {      // (descriptors based at the block's first column, bounded at the tensor's end: lds_dma.h)
            const __amdgpu_buffer_rsrc_t rs = rsrc_from(a.m[m].s, c0 * (G * 2u), (unsigned)(a.N * a.sh * 2));
            dma_piece_default(L::SIDE_S + m * L::SIDE_S_BYTES + p * 1024u, voff + (unsigned)p * 1024u, rs, 0u);
        } else {
            const __amdgpu_buffer_rsrc_t rz = rsrc_from(a.m[m].z, c0 * (ZW * 4u), (unsigned)(a.N * a.pzh * 4));
            dma_piece_default(L::SIDE_Z + m * L::SIDE_Z_BYTES + (p - (int)L::NS_S) * 1024u, voff + (unsigned)(p - (int)L::NS_S) * 1024u, rz, 0u);
        }
    }
    block_barrier_lds();      // the x loads are queued on this CU in front of every weight piece (the path returns in order)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mat ? arg_w1 : arg_w0), 0, (int)wbytes, 0x00020000);
    const unsigned ring = L::RING + (unsigned)wave * (D * 1024u);
    const unsigned soff0 = (c0 + ((unsigned)wave >> 1)) * CB;             // piece k = TS i + ks: soff0 + i * 8 columns + ks * 1024
    const bool odd = ((c0 + ((unsigned)wave >> 1)) & 1u) != 0u;           // (a wave's columns are 8 apart: one parity)
    const unsigned voff_half = (lane & 31u) * 16u;
    auto issue2 = [&](int i, int ks) {                 // piece ks of unit i (ks a constant at every call site)
        const unsigned dst = ring + (unsigned)((TS * i + ks) & (D - 1)) * 1024u, so = soff0 + (unsigned)i * (8u * CB) + (unsigned)ks * 1024u;
        if (TS == 3 && ks == 2) { if (upper == odd) dma_piece(dst, voff_half, rw, so); }   // 32 lanes: the LDS address follows the LANE, not the offset
        else dma_piece(dst, voff, rw, so);
    };
#pragma unroll
    for (int k = 0; k < D; k++)                        // the wave sends its whole ring at entry
        if (k < npieces) issue2(k / TS, k % TS);

    // ---- x chain (gemv_q4_body's staging, one 8-half chunk per thread of waves 0..7)
    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    float* tot = reinterpret_cast<float*>(smem + L::TOT);
    if (npieces >= D) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xraw), "+v"(wraw) : "n"(D) : "memory");   // all but the weight pieces
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xraw), "+v"(wraw) : : "memory");                          // (a narrow matrix: fewer were issued)
    if (NORM) {
        if (tid < (unsigned)(TS * 256)) part[tid] = stager ? sumsq8(xraw, 0.f) : 0.f;      // (K = 5120: entries 640 .. 767 are the zero padding of the canonical sum)
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
    u32x4 X[TS][4];
    float corr[TS];
#pragma unroll
    for (int ks = 0; ks < TS; ks++) {
        const unsigned lu = (TS == 3 && ks == 2) ? (lane & 31u) : lane;      // half slot: both halves of the wave hold units 0-31
#pragma unroll
        for (int d = 0; d < 4; d++) X[ks][d] = xs[((ks * 4 + d) << 6) + lu];
        corr[ks] = sx[ks * 64 + lu];
    }
    const unsigned char* wbase = smem + ring + lane * 16u;
    const unsigned char* sbase = smem + L::SIDE_S + mat * L::SIDE_S_BYTES + ((unsigned)(wave >> 1) * G + (lane >> 2)) * 2u;   // + i * 8 columns * 64 (80) B
    const unsigned char* zbase = smem + L::SIDE_Z + mat * L::SIDE_Z_BYTES + ((unsigned)(wave >> 1) * ZW + (lane >> 5)) * 4u;    // + i * 8 columns * 16 (20) B
    const unsigned zsh = ((lane >> 2) & 7u) * 4u;

    for (int g4 = 0; g4 * 4 < nu; g4++) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = g4 * 4 + r;
            if (i < nu) {
                float c = 0.f;
#pragma unroll
                for (int ks = 0; ks < TS; ks++) {
                    const int j = TS * i + ks;
                    constexpr int DM1 = D - 1;
                    const int e = (TS * r + ks) & DM1;                      // j % D (4 TS g4 is a multiple of D)
                    const bool hs = TS == 3 && ks == 2;                     // the half slot
                    // piece j has landed -- exact: the stream's last piece but one does not wait for the last (13B 580.3 -> 583.2, Mistral geometry
                    // 913.2 -> 914.4 tokens/s, tools/ab.py)
                    if (j + 1 < npieces) wait_vmcnt<DM1>(); else wait_vmcnt<0>();
                    const u32x4 w = *reinterpret_cast<const u32x4*>(wbase + e * 1024);
                    // scale and zero word of this lane's group: 16 ks + lane / 4, in the half slot 32 + (lane % 32) / 4
                    const uint16_t sc = *reinterpret_cast<const uint16_t*>(sbase + (unsigned)i * (8u * G * 2u) + (hs ? 64 - (int)(upper ? 16u : 0u) : ks * 32));
                    const unsigned zw = *reinterpret_cast<const unsigned*>(zbase + (unsigned)i * (8u * ZW * 4u) + (hs ? 16 - (int)(upper ? 4u : 0u) : ks * 8));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done: the entry may be refilled
                    if (j + D < npieces) issue2(i + (ks + D) / TS, (ks + D) % TS);
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
                        c += (upper == odd) ? v : 0.f;
                    } else {
                        c = __builtin_fmaf(h2f(sc), t, c);
                    }
                    // K = 5120: the piece's arithmetic stays HERE, in front of the next piece's wait. Left alone hipcc sinks a unit's 48 dot products behind
                    // the NEXT unit's three waits (and the last two units' behind the group's last wait): the wave sits in vmcnt with landed pieces
                    // unmultiplied. 13B -n 256: 565.6 -> 581.2 tokens/s (tools/ab.py, one call). At K = 4096 hipcc keeps a unit's 32 dot products behind
                    // the unit's own two waits, and pinning them per piece is neutral (Mistral geometry 911.8 vs 910.8): left to the compiler
                    if (TS == 3) asm volatile("" : "+v"(c));
                }
                cs[r] = c;
            }
        }
        const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row r: unit g4 * 4 + r
        const int row = lane >> 4;
        if ((lane & 15u) == 0 && g4 * 4 + row < nu) tot[wave + 16 * (g4 * 4 + row)] = total;   // [column][matrix] = unit index
    }
    block_barrier_lds();
    if ((int)tid < nc) {
        const float g = tot[2 * tid], u = tot[2 * tid + 1];
        float val = g;
        val *= 1.0f / (1.0f + expf(-val));              // gpu_kernels.h:271
        val *= u;                                       // :272
        a.out[0][c0 + tid] = f2h(val);
    }
}

// K = 5120 with PAIR units: a wave's unit is a pair of adjacent columns (c even, c + 1) of one matrix = FIVE full 1 KiB pieces -- c k-slot 0, c k-slot 1,
// c + 1 k-slot 0, c + 1 k-slot 1 and ONE piece for both columns' last 32 uint4 (lanes 0-31 fetch column c's, lanes 32-63 column c + 1's: the load and the
// single dequant-dot evaluation of gemv_q4.h's shared half slot) -- instead of two units of two pieces and a 512-byte one. A wave's stream is a latency
// chain of one memory round trip per two pieces whatever their size, so the 512-byte pieces cost bandwidth (a third of ffn_strip_kernel<..., 3>'s DMA
// instructions carried half a KiB: 6.5 TB/s in the steady state) and every sixth piece is saved. Same per-lane chains (k-slot 0, k-slot 1, half slot as a
// product and a sum on the column's own half of the wave), same reduction: bit for bit (tests/prof_cases.py). Needs an even number of columns per CU.
template <bool NORM>
__global__ void __launch_bounds__(STRIP_WAVES * 64) ffn_strip_pair_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w0, const void* arg_w1, const unsigned wbytes,
                                                                          const unsigned cbase, const GemvArgs a) {
    constexpr int TS = 3, D = 2, PPU = 5;              // k-slots of a column, ring depth, pieces per pair unit
    constexpr unsigned CB = 2560u, G = 40u, ZW = 5u;
    constexpr int NSTAGE = 10;
    using L = StripLds<D, TS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned c0 = blockIdx.x * cbase;            // (even)
    const int nc = (int)cbase;
    const int mat = wave & 1;
    const int npu = (nc - wave + 15) >> 4;             // this wave's pair units: pu = wave + 16 i of the block's nc (nc / 2 pairs x 2 matrices): pair wave / 2 + 8 i, matrix wave % 2
    const int npieces = PPU * npu;
    const unsigned voff = lane * 16u;
    const bool upper = lane >= 32u;
    const unsigned voff_pair = (upper ? CB : 0u) + (lane & 31u) * 16u;   // the shared piece: column c's tail below, column c + 1's above
    const bool stager = wave < NSTAGE;

    u32x4 xraw = {0u, 0u, 0u, 0u}, wraw = {0u, 0u, 0u, 0u};
    if (stager) {
        const u32x4* px = arg_x + tid;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xraw) : "v"(px) : "memory");
        if (NORM) {
            const u32x4* pw = arg_rms + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wraw) : "v"(pw) : "memory");
        }
    }
    constexpr int NSIDE = (int)(L::NS_S + L::NS_Z);
    if (wave < 2 * NSIDE) {
        const int m = wave >= NSIDE, p = wave - NSIDE * m;
        if (p < (int)L::NS_S) {      // (descriptors based at the block's first column, bounded at the tensor's end: lds_dma.h)
            const __amdgpu_buffer_rsrc_t rs = rsrc_from(a.m[m].s, c0 * (G * 2u), (unsigned)(a.N * a.sh * 2));
            dma_piece_default(L::SIDE_S + m * L::SIDE_S_BYTES + p * 1024u, voff + (unsigned)p * 1024u, rs, 0u);
        } else {
            const __amdgpu_buffer_rsrc_t rz = rsrc_from(a.m[m].z, c0 * (ZW * 4u), (unsigned)(a.N * a.pzh * 4));
            dma_piece_default(L::SIDE_Z + m * L::SIDE_Z_BYTES + (p - (int)L::NS_S) * 1024u, voff + (unsigned)(p - (int)L::NS_S) * 1024u, rz, 0u);
        }
    }
    block_barrier_lds();      // the x loads are queued on this CU in front of every weight piece
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mat ? arg_w1 : arg_w0), 0, (int)wbytes, 0x00020000);
    const unsigned ring = L::RING + (unsigned)wave * (D * 1024u);
    const unsigned cl0 = 2u * ((unsigned)wave >> 1);                      // first column of this wave's pair 0, local to the block; pair i: + 16 i
    const unsigned soff0 = (c0 + cl0) * CB;
    auto issue = [&](int i, int k) {                    // piece k of pair unit i (k a constant at every call site)
        const unsigned dst = ring + (unsigned)((PPU * i + k) & (D - 1)) * 1024u, so = soff0 + (unsigned)i * (16u * CB);
        if (k == 4) dma_piece(dst, voff_pair, rw, so + 2048u);
        else dma_piece(dst, voff, rw, so + (unsigned)(k >> 1) * CB + (unsigned)(k & 1) * 1024u);
    };
    if (npieces > 0) { issue(0, 0); issue(0, 1); }

    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    float* tot = reinterpret_cast<float*>(smem + L::TOT);
    if (npieces > 0) asm volatile("s_waitcnt vmcnt(2)" : "+v"(xraw), "+v"(wraw) : : "memory");   // all but the weight pieces
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xraw), "+v"(wraw) : : "memory");
    if (NORM) {
        if (tid < (unsigned)(TS * 256)) part[tid] = stager ? sumsq8(xraw, 0.f) : 0.f;
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
        cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);
        const unsigned j = tid >> 2, d = tid & 3u;
        xs[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
        if (d == 0) sx[j] = cb * -9.5367431640625e-07f;
    }
    block_barrier_lds();                               // x staged; side data landed
    u32x4 X[TS][4];
    float corr[TS];
#pragma unroll
    for (int ks = 0; ks < TS; ks++) {
        const unsigned lu = ks == 2 ? (lane & 31u) : lane;
#pragma unroll
        for (int d = 0; d < 4; d++) X[ks][d] = xs[((ks * 4 + d) << 6) + lu];
        corr[ks] = sx[ks * 64 + lu];
    }
    const unsigned char* wbase = smem + ring + lane * 16u;
    const unsigned char* sbase = smem + L::SIDE_S + mat * L::SIDE_S_BYTES + (cl0 * G + (lane >> 2)) * 2u;    // + i * 16 columns * 80 B, + 80 for the pair's second column
    const unsigned char* zbase = smem + L::SIDE_Z + mat * L::SIDE_Z_BYTES + (cl0 * ZW + (lane >> 5)) * 4u;   // + i * 16 columns * 20 B, + 20
    const unsigned zsh = ((lane >> 2) & 7u) * 4u;

    for (int g2 = 0; g2 * 2 < npu; g2++) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};             // pair 2 g2: columns c, c + 1; pair 2 g2 + 1: columns c, c + 1
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int i = g2 * 2 + r;
            if (i < npu) {
                float cA = 0.f, cB = 0.f;
#pragma unroll
                for (int k = 0; k < PPU; k++) {
                    const int j = PPU * i + k;
                    const int e = (PPU * r + k) & (D - 1);                  // j % D (two pair units are ten pieces)
                    if (j + 1 < npieces) wait_vmcnt<1>(); else wait_vmcnt<0>();   // piece j has landed
                    const u32x4 w = *reinterpret_cast<const u32x4*>(wbase + e * 1024);
                    const int ks = k == 4 ? 2 : (k & 1);
                    const unsigned colsel = k == 4 ? (upper ? 1u : 0u) : (unsigned)(k >> 1);   // which column of the pair this lane works for
                    // scale and zero word of this lane's group: 16 ks + lane / 4, in the shared piece 32 + (lane % 32) / 4 of the lane's own column
                    const uint16_t sc = *reinterpret_cast<const uint16_t*>(sbase + ((unsigned)i * 16u + colsel) * (G * 2u) + (k == 4 ? 64 - (int)(upper ? 16u : 0u) : ks * 32));
                    const unsigned zw = *reinterpret_cast<const unsigned*>(zbase + ((unsigned)i * 16u + colsel) * (ZW * 4u) + (k == 4 ? 16 - (int)(upper ? 4u : 0u) : ks * 8));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done: the entry may be refilled
                    if (j + D < npieces) issue(i + (k + D) / PPU, (k + D) % PPU);
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
                    if (k == 4) {        // gemv_q4.h's shared half slot: a product and a sum, each half of the wave for its own column
                        const float v = h2f(sc) * t;
                        cA += upper ? 0.f : v;
                        cB += upper ? v : 0.f;
                    } else if (k < 2) {
                        cA = __builtin_fmaf(h2f(sc), t, cA);
                    } else {
                        cB = __builtin_fmaf(h2f(sc), t, cB);
                    }
                    asm volatile("" : "+v"(cA), "+v"(cB));                  // the piece's arithmetic stays in front of the next piece's wait
                }
                cs[2 * r] = cA;
                cs[2 * r + 1] = cB;
            }
        }
        const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row: pair (2 g2 + row / 2), column row % 2
        const int row = lane >> 4;
        if ((lane & 15u) == 0 && g2 * 2 + (row >> 1) < npu)
            tot[2 * (cl0 + 16u * (unsigned)(g2 * 2 + (row >> 1)) + (unsigned)(row & 1)) + mat] = total;   // [column][matrix]
    }
    block_barrier_lds();
    if ((int)tid < nc) {
        const float g = tot[2 * tid], u = tot[2 * tid + 1];
        float val = g;
        val *= 1.0f / (1.0f + expf(-val));              // gpu_kernels.h:271
        val *= u;                                       // :272
        a.out[0][c0 + tid] = f2h(val);
    }
}

// Shapes: K = 4096 (two 1 KiB pieces per column), 16 .. 56 columns per CU, on a stream that may use every CU (the blocks do not wait for
// each other: on a CU-masked stream the form would be correct, only slow). Strips run from STRIP_MIN_COLS columns per CU on, where they measure
// faster than the wave-owned kernel (tools/lab/sweep_strips.py, per launch in a graph, wave-owned | strips: 16 columns per CU 4.92 | 5.39 us, 20: 5.88 | 5.92,
// 24: 6.25 | 6.46, 28: 7.12 | 6.98, 32: 7.78 | 7.75, 36: 8.57 | 8.23, 40: 9.15 | 9.01, 43 (Llama-2-7B): 10.12 | 9.69, 46: 10.08 | 9.72; the Mistral /
// Llama-3 hidden size 14336: 12.4 -> 11.45 us, 885 -> 902 tokens/s on the Mistral-7B geometry; Llama-2-7B -n 256 982.7 -> 990.8 tokens/s against a build
// with the threshold at 49). g_gemv_form (q4_internal.h) is GEMV_PRODUCT in the shipped library; the profiling library's knob 11 sets the others.
#ifdef Q4_STRIP_MIN_COLS
constexpr int STRIP_MIN_COLS = Q4_STRIP_MIN_COLS;   // (make exp EXPFLAGS_GEMV=-DQ4_STRIP_MIN_COLS=49: the wave-owned kernel for Llama-2-7B's gate/up, for the A/B)
#else
constexpr int STRIP_MIN_COLS = 36;
#endif
static inline bool strip_k5120(const GemvArgs& a) { return a.K == 5120 && a.pw4 == 160 && a.sh == 40 && a.pzh == 5; }
static bool ffn_strip_shape(const GemvArgs& a) {
    const int nb = cu_count();
    // K = 4096, or K = 5120 (13B) where the wave-owned kernel would run its shared half slot, whose arithmetic the strips repeat
    const bool k = (a.K == 4096 && a.pw4 == 128 && a.sh == 32 && a.pzh == 4) || (strip_k5120(a) && a.nslots == 3 && half_tail(a));
    return k && a.N / nb >= 16 && divUp(a.N, nb) <= STRIP_NCMAX && stream_cu_count() == nb;
}
static bool ffn_strip_covers(const GemvArgs& a) {
    if (g_gemv_form != GEMV_PRODUCT && g_gemv_form != GEMV_PRODUCT_NO_DOWN_STRIPS && g_gemv_form != GEMV_K5120_COLUMN_UNITS) return false;   // (GEMV_WAVE_OWNED; the laboratory forms answer through g_lab first)
    return a.N / cu_count() >= STRIP_MIN_COLS && g_ablate == 0 && ffn_strip_shape(a);
}
template <bool NORM, int TS>
static int launch_strip(const GemvArgs& a) {
    constexpr size_t smem = StripLds<2, TS>::BYTES;
    static_assert(smem <= 64 * 1024, "54 KiB: no opt-in, nothing that a graph capture could not record");
    const unsigned nb = (unsigned)cu_count();
    Q4_LAUNCH((ffn_strip_kernel<NORM, TS>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
              (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), (unsigned)a.N / nb, (unsigned)a.N % nb, a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
// K = 5120 as pair units: every CU the same even number of columns
static bool strip_pairs(const GemvArgs& a) { return strip_k5120(a) && a.N % (2 * cu_count()) == 0; }
static int launch_strip_pair(const GemvArgs& a) {
    constexpr size_t smem = StripLds<2, 3>::BYTES;
    static_assert(smem <= 64 * 1024, "no opt-in");
    const unsigned nb = (unsigned)cu_count();
    if (a.rms_w)
        Q4_LAUNCH((ffn_strip_pair_kernel<true>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
                  (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), (unsigned)a.N / nb, a);
    else
        Q4_LAUNCH((ffn_strip_pair_kernel<false>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
                  (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), (unsigned)a.N / nb, a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
static int launch_ffn_strip(const GemvArgs& a) {
    if (g_gemv_form != GEMV_K5120_COLUMN_UNITS && strip_pairs(a)) return launch_strip_pair(a);   // (the column-unit form of K = 5120 otherwise: uneven splits, and the A/B)
    if (strip_k5120(a)) return a.rms_w ? launch_strip<true, 3>(a) : launch_strip<false, 3>(a);
    return a.rms_w ? launch_strip<true, 2>(a) : launch_strip<false, 2>(a);
}

}  // namespace q4
