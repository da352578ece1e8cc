// gemv_strip_down.h -- the long-K plain GEMV (mat_vec_kernel_int4, gpu_kernels.h:213-240, with its `accum` epilogue) as strips, for the shape where
// gemv_q4.h's K-split kernel leaves CUs uneven: the 13B down projection (K = 13824, N = 5120: 640 four-wave blocks = 2.5 per CU, the launch pays
// for three). Same scheme as gemv_strip.h -- every wave streams its own units with `buffer_load_dwordx4 ... nt lds` into a private ring of two
// 1 KiB pieces, waits with vmcnt, reads back, re-issues, multiplies -- with these differences:
//   * one sixteen-wave block per CU, twenty columns per block at N = 5120: a unit is (column, k-part), two k-parts per column as in gemv_q4.h
//     (KS = 2: part p owns uint4 units [p * ku, (p + 1) * ku), ku = 216), 40 units = three or two per wave (two ten-wave blocks per CU, two
//     units per wave, were measured first: at the kernel's 119 VGPRs only one of them is resident, 11.8 us; held to 96 VGPRs it spills: 17 us);
//   * a unit is SH - 1 = 3 full pieces and a last one of ku - 192 = 24 uint4, which -- gemv_q4.h's shared half slot -- the lower half of the wave
//     takes for even columns and the upper half for odd ones, its term added as a product and a sum (not an fma);
//   * x (K halves = 27 KiB, no norm) is staged once per block in the permuted layout and read per piece (it does not fit registers);
//   * epilogue: column total = part 0 + part 1 (that order), `accum` adds the fp16 residual, one rounding.
// Same arithmetic in the same order as gemv_q4_kernel<MODE_PLAIN, 4, 4, false, 0, 2, true>: bit for bit (tests/prof_cases.py).
#pragma once
#include "gemv_strip.h"

namespace q4 {

constexpr int SD_WAVES = 16, SD_NCMAX = 24, SD_ROWS = 7;
   // 7 rows of 64 uint4 units: K <= 14336
struct StripDownLds {
    static constexpr unsigned RING = 0;                                 // [16 waves][2] x 1 KiB
    static constexpr unsigned SIDE_S = RING + SD_WAVES * 2048u;         // 6 KiB: 24 columns x 112 groups x 2 B = 5376 (13B: 20 x 216)
    static constexpr unsigned SIDE_Z = SIDE_S + 6144u;                  // 2 KiB: 24 columns x 14 words x 4 B = 1344 (13B: 20 x 56)
    static constexpr unsigned XS = SIDE_Z + 2048u;                      // [7][4][64] x 16 B permuted x
    static constexpr unsigned SX = XS + SD_ROWS * 4096u;                // [7][64] -(sum of the 32 x) * 2^-20
    static constexpr unsigned TOT = SX + SD_ROWS * 256u;                // [NCMAX][2] part totals
    static constexpr unsigned BYTES = TOT + SD_NCMAX * 8u;
};

// SH = slots of a k-part; SHARED: the last one holds at most 32 units and is shared between the columns of a pair (gemv_q4.h's HALF form; 13B),
// else it is an ordinary slot whose lanes past the part's end add nothing (7B: 172 = 64 + 64 + 44)
template <int SH, bool SHARED>
__global__ void __launch_bounds__(SD_WAVES * 64) down_strip_kernel(const u32x4* __restrict__ arg_x, const void* arg_w, const unsigned wbytes, const unsigned cbase,
                                                                 const unsigned crem, const GemvArgs a) {
    using L = StripDownLds;
    constexpr int PPU = SH;                            // DMA pieces per unit
    static_assert(SH >= 2 && SH <= 4, "a unit is SH pieces; ring entry of piece ks of unit i = (SH i + ks) % 2, i a constant at every site");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned c0 = blockIdx.x * cbase + (blockIdx.x < crem ? blockIdx.x : crem);
    const int nc = (int)(cbase + (blockIdx.x < crem ? 1u : 0u));
    const int part_id = wave & 1;                      // unit u = wave + 16 i: column c0 + u / 2, k-part u % 2 = wave % 2
    const int nu = (2 * nc - wave + SD_WAVES - 1) / SD_WAVES;
    const int npieces = PPU * nu;
    const unsigned voff = lane * 16u, lu = lane & 31u;
    const bool upper = lane >= 32u;
    const unsigned ku = (unsigned)a.ku, ubase = (unsigned)part_id * ku;
    const unsigned uend = ubase + ku < (unsigned)a.pw4 ? ubase + ku : (unsigned)a.pw4;
    const unsigned nchunks = (unsigned)a.K >> 3;
    const unsigned colbytes = (unsigned)a.pw4 * 16u;

    // ---- x first (the address arrives with the wave when the build preloads kernel arguments), then the side data, then the ring
    u32x4 xraw[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const unsigned u = tid + (unsigned)i * (SD_WAVES * 64u);
        const u32x4* px = arg_x + (u < nchunks ? u : nchunks - 1u);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xraw[i]) : "v"(px) : "memory");
    }
    if (wave < 8) {                                    // scales (6 KiB) and zeros (2 KiB) of the block's columns
        if (wave < 6) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_from(a.m[0].s, c0 * (unsigned)a.sh * 2u, (unsigned)(a.N * a.sh * 2));   // (bounded at the tensor's end: lds_dma.h)
            dma_piece_default(L::SIDE_S + (unsigned)wave * 1024u, voff + (unsigned)wave * 1024u, rs, 0u);
        } else {
            const __amdgpu_buffer_rsrc_t rz = rsrc_from(a.m[0].z, c0 * (unsigned)a.pzh * 4u, (unsigned)(a.N * a.pzh * 4));
            dma_piece_default(L::SIDE_Z + (unsigned)(wave - 6) * 1024u, voff + (unsigned)(wave - 6) * 1024u, rz, 0u);
        }
    }
    block_barrier_lds();      // the x loads are queued on this CU in front of every weight piece
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(arg_w), 0, (int)wbytes, 0x00020000);
    const unsigned ring = L::RING + (unsigned)wave * 2048u;
    // piece ks of unit i: column c0 + wave / 2 + 8 i, bytes (ubase + 64 ks) * 16 onwards; the last piece: `tail` uint4 on one half of the wave
    const unsigned col0 = c0 + ((unsigned)wave >> 1);
    const unsigned tail = ku - (unsigned)(SH - 1) * 64u;                 // uint4 units in the shared slot (<= 32)
    auto issue2 = [&](int i, int ks) {
        const unsigned col = col0 + 8u * (unsigned)i;
        const unsigned dst = ring + (unsigned)((PPU * i + ks) & 1) * 1024u, so = col * colbytes + (ubase + 64u * (unsigned)ks) * 16u;
        if (ks == SH - 1) {
            if (SHARED) { if (upper == ((col & 1u) != 0u) && lu < tail && ubase + 64u * (unsigned)ks + lu < uend) dma_piece(dst, lu * 16u, rw, so); }
            else { if (lane < tail && ubase + 64u * (unsigned)ks + lane < uend) dma_piece(dst, voff, rw, so); }
        } else dma_piece(dst, voff, rw, so);
    };
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (k < npieces) issue2(0, k);

    // ---- x staging: gemv_q4_body's, two chunks per thread; no norm on this path (llama2_q4.cu:331)
    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* tot = reinterpret_cast<float*>(smem + L::TOT);
    if (npieces >= 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(xraw[0]), "+v"(xraw[1]) : : "memory");   // all but the two weight pieces
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xraw[0]), "+v"(xraw[1]) : : "memory");
    {
        const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const unsigned u = tid + (unsigned)i * (SD_WAVES * 64u);
            u32x4 v = xraw[i];
            if (u >= nchunks) v = (u32x4){0u, 0u, 0u, 0u};
            v = q4_signed_x(v, q4_stage_sign_bits(tid));      // odd units are staged negated (gemv_q4.h, q4_stage_sign_bits)
            const u32x4 pv = permute_x8(v);
            float cb = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
            cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);   // quad sum: the 32 inputs of one uint4 unit
            const unsigned j = u >> 2, d = u & 3u;
            if (u < (unsigned)(SD_ROWS * 256)) {
                xs[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
                if (d == 0) sx[j] = cb * -9.5367431640625e-07f;     // -(sum x) * 2^-20
            }
        }
    }
    block_barrier_lds();                               // x staged; side data landed (its issuers passed the vmcnt wait above)
    const unsigned char* wbase = smem + ring + lane * 16u;
    const unsigned char* sside = smem + L::SIDE_S;
    const unsigned char* zside = smem + L::SIDE_Z;

    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; i++) {                      // at most 2 * SD_NCMAX / SD_WAVES = 3 units per wave
        if (i < nu) {
            const unsigned lc = ((unsigned)wave >> 1) + 8u * (unsigned)i;          // column inside the block
            const bool odd = ((c0 + lc) & 1u) != 0u;
            float c = 0.f;
#pragma unroll
            for (int ks = 0; ks < SH; ks++) {
                const int j = PPU * i + ks;
                const bool last = ks == SH - 1, hs = SHARED && last;
                if (j + 1 < npieces) wait_vmcnt<1>(); else wait_vmcnt<0>();        // piece j has landed (the last piece but one does not wait for the last)
                const unsigned unit = ubase + 64u * (unsigned)ks + (hs ? lu : lane);   // the uint4 unit this lane multiplies
                const bool live = !last || (SHARED ? (upper == odd && lu < tail && unit < uend) : (lane < tail && unit < uend));
                const unsigned uj = unit < uend ? unit : uend - 1u;
                const u32x4 w = *reinterpret_cast<const u32x4*>(wbase + ((PPU * i + ks) & 1) * 1024);
                const uint16_t sc = *reinterpret_cast<const uint16_t*>(sside + (lc * (unsigned)a.sh + (uj >> 2)) * 2u);
                const unsigned zw = *reinterpret_cast<const unsigned*>(zside + (lc * (unsigned)a.pzh + (uj >> 5)) * 4u);
                const unsigned xrow = ((uj >> 6) << 8) + (uj & 63u);
                u32x4 X[4];
#pragma unroll
                for (int d = 0; d < 4; d++) X[d] = xs[xrow + (d << 6)];
                const float corr = sx[uj];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the reads are done: the entry may be refilled
                if (j + 2 < npieces) issue2(i + (ks + 2) / PPU, (ks + 2) % PPU);
                float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned ww = w[d];
                    const unsigned tt = ww >> 8;
                    acc_e = __builtin_amdgcn_fdot2(as_h2(ww & 0x000F000Fu), as_h2(X[d][0]), acc_e, false);
                    acc_o = __builtin_amdgcn_fdot2(as_h2(ww & 0x00F000F0u), as_h2(X[d][1]), acc_o, false);
                    acc_e = __builtin_amdgcn_fdot2(as_h2(tt & 0x000F000Fu), as_h2(X[d][2]), acc_e, false);
                    acc_o = __builtin_amdgcn_fdot2(as_h2(tt & 0x00F000F0u), as_h2(X[d][3]), acc_o, false);
                }
                const unsigned zsh = ((uj >> 2) & 7u) * 4u;
                const float zf = (float)((zw >> zsh) & 0xFu);
                float t = __builtin_fmaf(acc_e, 16.f, acc_o);
                t = __builtin_fmaf(zf, corr, t);
                if (hs) {           // the shared slot: a product and a sum, only on the half of the wave (and the lanes) that hold this column's units
                    const float v = h2f(sc) * t;
                    c += live ? v : 0.f;
                } else if (last) {  // an ordinary last slot: its lanes past the part's end multiply zero inputs in gemv_q4.h, fma(s, 0, c) = c
                    const float f = __builtin_fmaf(h2f(sc), t, c);
                    c = live ? f : c;
                } else {
                    c = __builtin_fmaf(h2f(sc), t, c);
                }
                // the piece's arithmetic stays HERE, in front of the next piece's wait: left alone hipcc sinks it behind later waits (the unrolled pieces
                // sit in registers meanwhile and the wave idles in vmcnt with multiplications pending). 13B -n 256: 565.6 -> 569.3 tokens/s
                asm volatile("" : "+v"(c));
            }
            cs[i] = c;
        }
    }
    {
        const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row r: unit r of this wave
        const int row = lane >> 4;
        if ((lane & 15u) == 0 && row < nu) tot[wave + SD_WAVES * row] = total;      // [column][part] = unit index
    }
    block_barrier_lds();
    if ((int)tid < nc) {
        float r = tot[2 * tid];
        r += tot[2 * tid + 1];                          // fixed order: k-parts from the lowest up
        q4_half* out = a.out[0];
        const unsigned n = c0 + tid;
        if (a.accum) r += h2f(out[n]);                  // gpu_kernels.h:229-230
        out[n] = f2h(r);                                // :231
    }
}

// the shapes: the K-split kernel (two k-parts, ku uint4 each) with up to four slots per part and at most seven 64-unit rows of x (K <= 14336), no KV
// addressing, on a stream that may use every CU; 8 .. 24 columns per block. Llama-2-13B: K = 13824, four slots, the last shared (24 units): 9.83 ->
// 8.64 us per launch, the product's choice. Llama-2-7B: K = 11008, three slots, the last an ordinary one of 44 units: bit-identical as well, and
// SLOWER than the K-split kernel's 512 blocks = exactly two per CU (962 -> 936 tokens/s): only under the profiling build's GEMV_STRIPS_EVERYWHERE
// (which forces strips wherever the shape is covered).
static bool down_strip_covers(const GemvArgs& a, bool shared) {
    const int nb = cu_count();
    if (g_gemv_form == GEMV_PRODUCT || g_gemv_form == GEMV_K5120_COLUMN_UNITS) {
        // the product's choice: the shared-slot form (four slots per k-part, K = 13824 / 14336). It was taken only where the K-split kernel's grid (8 columns
        // per block) leaves the CUs uneven -- 13B: 640 blocks = 2.5 per CU, the launch pays for three; strips 536 -> 546 tokens/s -- because on an even grid
        // the K-split kernel was faster (Mistral geometry, K = 14336, 512 blocks: strips 899 -> 890 tokens/s). With the strips' kernel arguments preloaded
        // (csrc/Makefile GEMVFLAGS), their arithmetic pinned per piece and the exact tail wait they win there too: 7.63 -> 7.47 us per launch, 908.3 -> 914.1
        // tokens/s. The three-slot form (Llama-2-7B, K = 11008) still loses (6.02 vs 6.30 us) and stays a profiling knob.
        if (!shared) return false;
    } else if (g_gemv_form != GEMV_STRIPS_EVERYWHERE) return false;
    const int sh = divUp(a.ku, 64);
    const bool shape = shared ? (sh == 4 && a.ku - (sh - 1) * 64 <= 32) : (sh == 3 && a.ku - (sh - 1) * 64 > 32);
    return a.nslots >= 5 && shape && a.ku * 2 >= a.pw4 && divUp(a.pw4, 64) <= SD_ROWS && a.loff == -1 && a.rms_w == nullptr &&
           a.N / nb >= 8 && divUp(a.N, nb) <= SD_NCMAX && g_ablate == 0 && stream_cu_count() == nb;
}
// 70 KiB of LDS: the opt-in is not a stream operation -- build_transformer makes it for the model (q4_runtime.hip), a stand-alone call at its first launch
int down_strip_prepare() {
    const int rc = lds_opt_in((const void*)down_strip_kernel<4, true>, StripDownLds::BYTES);      // (once per device)
    return rc ? rc : lds_opt_in((const void*)down_strip_kernel<3, false>, StripDownLds::BYTES);
}
template <int SH, bool SHARED>
static int launch_down_strip(const GemvArgs& a) {
    { const int rc = down_strip_prepare(); if (rc) return rc; }
    const unsigned blocks = (unsigned)cu_count();
    Q4_LAUNCH((down_strip_kernel<SH, SHARED>), dim3(blocks), dim3(SD_WAVES * 64), StripDownLds::BYTES, reinterpret_cast<const u32x4*>(a.x), (const void*)a.m[0].w,
              (unsigned)(a.N * a.pw4 * 16), (unsigned)a.N / blocks, (unsigned)a.N % blocks, a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // namespace q4
