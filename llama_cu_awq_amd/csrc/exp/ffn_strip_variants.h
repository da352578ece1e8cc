// exp/ffn_strip_variants.h -- LABORATORY (libllama2_q4_prof.so only; the product never includes this file). The gate/up strips kernel of
// gemv_strip.h as it stood when its variants were measured (round 4; EXPERIMENTS.md "strips"): ring depth D = 2 / 4 / 8, and
//   MODE 0 ("plain"): a wave sends its whole ring at entry and re-issues an entry when it has read it: D pieces in flight per wave.
//   MODE 1 ("paced"): at most TWO pieces of a wave in flight whatever the depth of its ring (32 KiB per CU is what the CU's memory
//          pipe takes without stalling the issue; more in flight measured slower): an issue is preceded by vmcnt(1). The waves that do
//          not stage x fill their rings during the x chain, one piece per piece landed; the x chain synchronises through LDS counters
//          (a wave stalled in vmcnt must not hold a hardware barrier up).
//   MODE 2: MODE 1 with the dealing order rotated by eight waves: the waves that do not stage x take the longer share.
//   STAMPS: per-wave wall-clock stamps kept in LDS and written out at the end (tools/lab/timeline_strip.py).
// D = 2, MODE 0 without stamps is the product's kernel (gemv_strip.h, ffn_strip_kernel): tests/prof_cases.py holds every variant to its bits.
#pragma once
#include "../gemv_strip.h"
#include "lds_flags.h"

namespace q4 {

enum { SF_SS = 0, SF_STAGED = 1, SF_FAIL = 2 };
// stamps (tools/lab/timeline_strip.py; kept in LDS and written out at the end: a global store would count in vmcnt): [0] wave 0 entry,
// [1] its x landed, [2] sum of squares exchanged, [3] x staged, [4 + i] its unit i multiplied, [12] its totals written, [13] outputs
// stored; [16 + w] wave w entry, [32 + w] wave w's first piece read, [48 + w] wave w's last unit multiplied
#define SSTAMP(k) do { if (STAMPS && lane == 0) st[(k)] = wall_clock64(); } while (0)
template <bool NORM, int D, int MODE, bool STAMPS, int TS = 2>
__global__ void __launch_bounds__(STRIP_WAVES * 64) ffn_strip_variant_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w0, const void* arg_w1, const unsigned wbytes,
                                                                     const unsigned cbase, const unsigned crem, const GemvArgs a) {
    // the scalars the entry needs come first: built with -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave instead of
    // through a scalar load from the kernel-argument segment (tools/lab/timeline_strip.py: "x landed")
    static_assert(D == 2 || D == 4 || D == 8, "ring entries of a unit's pieces are compile-time constants");
    static_assert(TS == 2 || (TS == 3 && MODE == 0 && D <= 4), "K = 5120: the plain form, 12 pieces per four units a multiple of the ring");
    constexpr bool PACED = MODE >= 1;
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
    const int wv = MODE == 2 ? ((wave + 8) & 15) : wave;   // position in the dealing order
    const int nu = (2 * nc - wv + 15) >> 4;            // this wave's units: u = wv + 16 i, column c0 + u / 2, matrix u % 2
    const int npieces = TS * nu;
    const unsigned voff = lane * 16u;
    const bool stager = wave < NSTAGE;
    // K = 5120: a column's third piece is 512 bytes, 32 lanes. As in gemv_q4.h's shared half slot the lower half of the wave takes it
    // for the first column of a pair (even), the upper half for the second (odd): the same lanes add the same terms
    const bool upper = lane >= 32u;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(smem + L::STAMP);
    unsigned* flags = reinterpret_cast<unsigned*>(smem + L::FLAGS);
    SSTAMP(16 + wave);
    if (wave == 0) SSTAMP(0);
    if (PACED && wave == 15 && lane < 16u) flags[lane] = 0u;

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
    block_barrier_lds();      // the x loads are queued on this CU in front of every weight piece (the path returns in order); flags are zero
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mat ? arg_w1 : arg_w0), 0, (int)wbytes, 0x00020000);
    const unsigned ring = L::RING + (unsigned)wave * (D * 1024u);
    const unsigned soff0 = (c0 + ((unsigned)wv >> 1)) * CB;               // piece k = TS i + ks: soff0 + i * 8 columns + ks * 1024
    const bool odd = ((c0 + ((unsigned)wv >> 1)) & 1u) != 0u;             // (a wave's columns are 8 apart: one parity)
    const unsigned voff_half = (lane & 31u) * 16u;
    auto issue2 = [&](int i, int ks) {                 // piece ks of unit i (ks a constant at every call site)
        const unsigned dst = ring + (unsigned)((TS * i + ks) & (D - 1)) * 1024u, so = soff0 + (unsigned)i * (8u * CB) + (unsigned)ks * 1024u;
        if (TS == 3 && ks == 2) { if (upper == odd) dma_piece(dst, voff_half, rw, so); }   // 32 lanes: the LDS address follows the LANE, not the offset
        else dma_piece(dst, voff, rw, so);
    };
    auto issue = [&](int k) { static_assert(TS == 2 || !PACED, "linear piece numbers: two pieces per unit"); issue2(k >> 1, k & 1); };
    constexpr int D0 = PACED ? 2 : D;                  // pieces a wave sends at entry
    int I = npieces < D0 ? npieces : D0;               // pieces issued so far
#pragma unroll
    for (int k = 0; k < D0; k++)
        if (k < npieces) issue2(k / TS, k % TS);

    // ---- x chain (gemv_q4_body's staging, one 8-half chunk per thread of waves 0..7)
    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    float* tot = reinterpret_cast<float*>(smem + L::TOT);
    if (stager || !PACED) {
        if (npieces >= D0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xraw), "+v"(wraw) : "n"(D0) : "memory");   // all but the weight pieces
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xraw), "+v"(wraw) : : "memory");                            // (a narrow matrix: fewer were issued)
    }
    if (wave == 0) SSTAMP(1);
    if (NORM) {
        if (tid < (unsigned)(TS * 256)) part[tid] = stager ? sumsq8(xraw, 0.f) : 0.f;      // (K = 5120: entries 640 .. 767 are the zero padding of the canonical sum)
        if (!PACED) block_barrier_lds();
        else if (stager) { lds_bump(&flags[SF_SS], lane); lds_wait_ge(&flags[SF_SS], 8u, &flags[SF_FAIL]); }
        if (wave == 0) SSTAMP(2);
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
    if (!PACED) block_barrier_lds();                   // x staged; side data landed (its issuers passed the vmcnt wait above)
    else {
        if (stager) lds_bump(&flags[SF_STAGED], lane);                      // a wave's LDS operations execute in order: its writes are in front
        else {
            // the other eight waves fill their rings meanwhile, one piece for every piece that lands (never more than two in flight);
            // waves 8 and 9 carry side pieces: theirs are older than their ring, so "at most one outstanding" covers them
            bool told = wave >= 10;
            while (I < npieces && I < D) {
                wait_vmcnt<1>();
                if (!told) { lds_bump(&flags[SF_STAGED], lane); told = true; }
                issue(I);
                I++;
                if (lds_peek(&flags[SF_STAGED]) >= 10u) break;
            }
            if (!told) { wait_vmcnt<1>(); lds_bump(&flags[SF_STAGED], lane); }
        }
        lds_wait_ge(&flags[SF_STAGED], 10u, &flags[SF_FAIL]);
    }
    if (wave == 0) SSTAMP(3);
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
    const unsigned char* sbase = smem + L::SIDE_S + mat * L::SIDE_S_BYTES + ((unsigned)(wv >> 1) * G + (lane >> 2)) * 2u;   // + i * 8 columns * 64 (80) B
    const unsigned char* zbase = smem + L::SIDE_Z + mat * L::SIDE_Z_BYTES + ((unsigned)(wv >> 1) * ZW + (lane >> 5)) * 4u;    // + i * 8 columns * 16 (20) B
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
                    if (!PACED) {
                        // piece j has landed. D = 2: exact -- the stream's last piece but one does not wait for the last (13B 580.3 -> 583.2, Mistral geometry
                        // 913.2 -> 914.4 tokens/s, tools/ab.py); deeper rings (profiling) drain at their last D pieces
                        if (D == 2 ? j + 1 < npieces : j + D < npieces) wait_vmcnt<DM1>(); else wait_vmcnt<0>();
                    } else {
                        if (j + 1 < npieces) wait_vmcnt<1>(); else wait_vmcnt<0>();     // pieces < I - 1 have landed, and I >= j + 2: piece j has
                        if (I < npieces && I - j < D) { issue(I); I++; }    // at most one was in flight: now two; the entry's last reader was piece I - D < j
                    }
                    if (j == 0) SSTAMP(32 + wave);
                    const u32x4 w = *reinterpret_cast<const u32x4*>(wbase + e * 1024);
                    // scale and zero word of this lane's group: 16 ks + lane / 4, in the half slot 32 + (lane % 32) / 4
                    const uint16_t sc = *reinterpret_cast<const uint16_t*>(sbase + (unsigned)i * (8u * G * 2u) + (hs ? 64 - (int)(upper ? 16u : 0u) : ks * 32));
                    const unsigned zw = *reinterpret_cast<const unsigned*>(zbase + (unsigned)i * (8u * ZW * 4u) + (hs ? 16 - (int)(upper ? 4u : 0u) : ks * 8));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done: the entry may be refilled
                    if (!PACED && j + D < npieces) issue2(i + (ks + D) / TS, (ks + D) % TS);
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
                if (STAMPS) { asm volatile("" : "+v"(c)); if (wave == 0 && i < 8) SSTAMP(4 + i); if (i == nu - 1) SSTAMP(48 + wave); }
            }
        }
        const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row r: unit g4 * 4 + r
        const int row = lane >> 4;
        if ((lane & 15u) == 0 && g4 * 4 + row < nu) tot[wv + 16 * (g4 * 4 + row)] = total;   // [column][matrix] = unit index
    }
    if (wave == 0) SSTAMP(12);
    block_barrier_lds();
    if ((int)tid < nc) {
        const float g = tot[2 * tid], u = tot[2 * tid + 1];
        float val = g;
        val *= 1.0f / (1.0f + expf(-val));              // gpu_kernels.h:271
        val *= u;                                       // :272
        a.out[0][c0 + tid] = (PACED && lds_peek(&flags[SF_FAIL]) != 0u) ? (uint16_t)0x7E00u : f2h(val);   // (NaN: a wait ran out)
    }
    if (STAMPS) {
        if (wave == 0) SSTAMP(13);
        block_barrier_lds();
        if (a.dbg && tid < 64u) a.dbg[(size_t)blockIdx.x * 64 + tid] = st[tid];
    }
}
#undef SSTAMP

template <bool NORM, int D, int MODE, bool STAMPS, int TS = 2>
static int launch_strip_variant(const GemvArgs& a) {
    constexpr size_t smem = StripLds<D, TS>::BYTES;
    // (the product's D = 2 form needs 54 KiB: no opt-in, nothing that a graph capture could not record)
    { const int rc = lds_opt_in((const void*)ffn_strip_variant_kernel<NORM, D, MODE, STAMPS, TS>, smem); if (rc) return rc; }
    const unsigned nb = (unsigned)cu_count();
    Q4_LAUNCH((ffn_strip_variant_kernel<NORM, D, MODE, STAMPS, TS>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
              (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), (unsigned)a.N / nb, (unsigned)a.N % nb, a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // namespace q4
