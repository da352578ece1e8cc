// exp/ffn_engine.h -- LABORATORY (libllama2_q4_prof.so only; knob 11 = 1..6). the fused gate/up GEMV at K = 4096 as a loader / consumer engine on LDS-DMA (MI355X_MICROARCH.md rows
// "ldsdma-fill", "nt-weights", "engine-vs-launches"). Same arithmetic as gemv_q4_kernel<MODE_FFN> (rmsnorm_kernel +
// ffn_matvec_silu_kernel, gpu_kernels.h:72-105, 256-275), bit for bit (tests/prof_cases.py compares the forms). The shipped library does
// not contain the engine: measured on MI355X (EXPERIMENTS.md notebook §9.12, profiles/r04_engine_records.txt) it lands the 47 MB in
// 7.0 us at 7.5 TB/s and still ends AFTER the wave-owned kernel -- 9.76 against 9.51-9.65 us per launch by rocprofv3 in one call, 937
// against 951 tokens/s as the token loop's gate/up launch -- its two consumer waves per SIMD do not keep up with the dequant (the strips
// form, four self-loading waves per SIMD, does).
//
// Idea: in gemv_q4.h the wave that loads a weight also multiplies it, so the depth of the prefetch is bounded by its VGPRs, no
// weight request goes out before the x chain of its block has been scheduled around, and every block re-stages x. Here one block
// per CU owns a contiguous range of column quads, and inside it
//   * ONE loader wave streams that range with `buffer_load_dwordx4 ... nt lds` (1 KiB per instruction, straight into an
//     8 x 16 KiB LDS ring; no VGPR ever holds a weight on the way in). It starts at t = 0 -- the x chain and the weight stream
//     overlap by construction -- and runs ahead of the consumers by the depth of the ring;
//   * EIGHT consumer waves stage x once per CU (fused rmsnorm, the canonical reduction of q4_device.h), keep their 32 registers of
//     permuted x for the whole kernel, and retire one ring slot together: wave w multiplies column w/2 of the quad for matrix w%2
//     (gate / up), both k-slots, with the denormal-nibble v_dot2c body of gemv_q4.h.
// A ring slot = one column quad = 4 columns x 2 matrices x K/8 bytes (16 KiB at K = 4096), contiguous per matrix in HBM
// (QWeight is column-major per output column: consecutive columns follow each other), so a fill is 16 fully coalesced
// 1 KiB pieces. Scales and zeros of the block's whole range (7 KiB) are fetched once, by the same loader, in front of the ring.
// (tools/lab/t_ldsdma.hip pins the instruction's semantics on gfx950: LDS address = M0 + immediate offset + lane * 16, M0 takes the
// full 160 KiB range.)
//
// Synchronisation inside the block (no s_barrier after the entry one: the loader must never wait for a consumer's pace):
//   landed   (LDS word) fills the loader KNOWS complete: after issuing fill j it waits vmcnt(16 * LAG) -- loads return in order,
//            so fills <= j - LAG have landed -- and stores j + 1 - LAG; at the end it drains LAG-1 .. 0. vmcnt is a 6-bit
//            counter, so a wave has at most 64 KiB of 1 KiB pieces in flight: LAG <= 3, and what the loader may have in flight
//            is exactly what it cannot know to have landed.
//   read[w]  (one LDS word per consumer wave) = slots wave w has read, stored by the wave once its ds_reads of a slot have been
//            executed; the loader refills slot j % 8 when every read[w] >= j - 7 (a single running total would not do: the waves
//            are not in lock step, four of them two slots ahead count like eight of them one slot ahead).
//   consumer-only barriers of the x chain: one LDS counter each.
#pragma once
#include "../gemv_strip.h"
#include "lds_flags.h"

namespace q4 {


constexpr int ENG_CONSUMERS = 8, ENG_RING = 8, ENG_NQMAX = 14;   // 14 quads per CU: hidden_dim up to 14336 on 256 CUs

template <int KSL>
struct EngLds {
    static constexpr unsigned SLOT = 8u * KSL * 1024u;                 // 4 columns x 2 matrices x KSL KiB
    static constexpr unsigned RING = 0;
    static constexpr unsigned SIDE_S_BYTES = ((ENG_NQMAX * 128u * KSL) + 1023u) & ~1023u;   // per matrix: quad = 4 cols x 16 KSL groups x 2 B
    static constexpr unsigned SIDE_S = RING + ENG_RING * SLOT;
    static constexpr unsigned SIDE_Z_BYTES = 1024u;                    // per matrix: quad = 4 cols x 2 KSL words x 4 B (<= 1 KiB for KSL <= 2)
    static constexpr unsigned SIDE_Z = SIDE_S + 2 * SIDE_S_BYTES;
    static constexpr unsigned XS = SIDE_Z + 2 * SIDE_Z_BYTES;          // [KSL][4][64] x 16 B permuted x
    static constexpr unsigned SX = XS + KSL * 4096u;                   // [KSL][64] -(sum of the 32 x) * 2^-20
    static constexpr unsigned PART = SX + KSL * 256u;                  // [KSL * 256] rmsnorm chunk partials
    static constexpr unsigned TOT = PART + KSL * 1024u;                // [NQMAX][8] column totals
    static constexpr unsigned FLAGS = TOT + ENG_NQMAX * 32u;           // landed, consumed, three consumer barriers
    static constexpr unsigned BYTES = FLAGS + 64u;
    static_assert(ENG_NQMAX * 32u * KSL <= SIDE_Z_BYTES, "zeros of the block's range: one DMA instruction per matrix");
};
enum { F_LANDED = 0, F_BAR0 = 2, F_BAR1 = 3, F_BAR2 = 4, F_FAIL = 5, F_READ0 = 8 };   // [F_READ0 + w]: slots consumer wave w has read

// the loader's wait for a ring slot: all ENG_CONSUMERS per-wave counts (one word each, lanes 0..7 read one apiece) >= target
__device__ __forceinline__ void lds_wait_all_ge(unsigned* p, unsigned target, unsigned lane, unsigned* fail) {
    for (unsigned n = 0;; n++) {
        const unsigned v = lane < (unsigned)ENG_CONSUMERS ? __hip_atomic_load(p + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xFFFFFFFFu;
        asm volatile("" ::: "memory");
        if (__ballot(v < target) == 0ull) break;
        if (n >= ENG_SPIN_LIMIT || ((n & 1023u) == 1023u && lds_peek(fail) != 0u)) { lds_post(fail, 1u, 0u); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// barrier among the consumer waves only (the loader keeps issuing)
__device__ __forceinline__ void consumer_barrier(unsigned* cnt, unsigned lane, unsigned* fail) {
    lds_bump(cnt, lane);
    lds_wait_ge(cnt, (unsigned)ENG_CONSUMERS, fail);
}

// wall-clock stamps (100 MHz, the same counter on every XCD) of one block's loader and of its consumer wave 0, 64 words per
// block: [0] loader entry, [1] side data issued, [2 + j] fill j known landed, [15] all landed; [16] consumer entry, [17] x staged,
// [18 + i] slot i seen landed, [31 + i] slot i multiplied, [44] totals exchanged, [45] outputs stored (tools/lab/timeline_engine.py).
// A stamp is a global store: it counts in the loader's vmcnt, so stamped runs of LAG = 1 stall on their own stamps.
#define ENG_STAMP(k) do { if (STAMPS && a.dbg && lane == 0) a.dbg[(size_t)blockIdx.x * 64 + (k)] = wall_clock64(); } while (0)

template <int KSL, bool NORM, int LAG, bool STAMPS, bool PF>
__global__ void __launch_bounds__((ENG_CONSUMERS + 1) * 64) ffn_engine_kernel(const GemvArgs a, const unsigned qbase, const unsigned qrem) {
    static_assert(KSL == 2, "x staging: one 8-half chunk per consumer thread (K = 4096)");
    static_assert(LAG >= 1 && 8 * KSL * LAG <= 48, "vmcnt is 6 bits");
    using L = EngLds<KSL>;
    constexpr int PIECES = 4 * KSL;              // 1 KiB pieces per matrix and quad
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* flags = reinterpret_cast<unsigned*>(smem + L::FLAGS);
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block b owns column quads [q0, q0 + nq): N / 4 quads dealt out evenly, the first qrem blocks take one more
    const unsigned q0 = blockIdx.x * qbase + (blockIdx.x < qrem ? blockIdx.x : qrem);
    const int nq = (int)(qbase + (blockIdx.x < qrem ? 1u : 0u));

    // The two roles part BEFORE any vector load: after a join hipcc would make the loader wait (vmcnt(0)) for the consumers' x loads
    // wherever it re-uses one of their registers. Both paths meet the entry barrier once (the hardware counts arrivals, not sites).
    if (wave == ENG_CONSUMERS) {
        if (lane < 16u) flags[lane] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // flags are zero, and the consumers' x loads are queued on this CU
        // ---------------------------------------------------------------- loader
        __builtin_amdgcn_s_setprio(3);
        ENG_STAMP(0);
        const unsigned voff = lane * 16u;
        __amdgpu_buffer_rsrc_t rw[2], rs[2], rz[2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            rw[m] = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].w, 0, a.N * a.pw4 * 16, 0x00020000);
            rz[m] = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].z, 0, a.N * a.pzh * 4, 0x00020000);
            rs[m] = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].s, 0, a.N * a.sh * 2, 0x00020000);
        }
        // scales and zeros of quads [q0, q0 + nq): contiguous per matrix; lanes past the tensor's end read nothing
#pragma unroll
        for (int m = 0; m < 2; m++) {
#pragma unroll
            for (unsigned i = 0; i < L::SIDE_S_BYTES / 1024u; i++)
                dma_piece_default(L::SIDE_S + m * L::SIDE_S_BYTES + i * 1024u, voff + i * 1024u, rsrc_from(a.m[m].s, q0 * (128u * KSL), (unsigned)(a.N * a.sh * 2)), 0u);
            dma_piece_default(L::SIDE_Z + m * L::SIDE_Z_BYTES, voff, rsrc_from(a.m[m].z, q0 * (32u * KSL), (unsigned)(a.N * a.pzh * 4)), 0u);
        }
        ENG_STAMP(1);
        for (int j = 0; j < nq; j++) {
            if (j >= ENG_RING) lds_wait_all_ge(&flags[F_READ0], (unsigned)(j - ENG_RING + 1), lane, &flags[F_FAIL]);   // every wave has read the slot's previous fill
            const unsigned slot = L::RING + (unsigned)(j % ENG_RING) * L::SLOT;
            const unsigned soff = (q0 + (unsigned)j) * (L::SLOT / 2u);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int p = 0; p < PIECES; p++)
                    dma_piece(slot + (unsigned)(m * PIECES + p) * 1024u, voff, rw[m], soff + (unsigned)p * 1024u);
            wait_vmcnt<2 * PIECES * LAG>();          // fills <= j - LAG have landed
            if (j + 1 - LAG > 0) { lds_post(&flags[F_LANDED], (unsigned)(j + 1 - LAG), lane); ENG_STAMP(2 + j - LAG); }
        }
        if (LAG > 2) { wait_vmcnt<2 * PIECES * 2>(); if (nq - 2 > 0) { lds_post(&flags[F_LANDED], (unsigned)(nq - 2), lane); ENG_STAMP(2 + nq - 3); } }
        if (LAG > 1) { wait_vmcnt<2 * PIECES * 1>(); if (nq - 1 > 0) { lds_post(&flags[F_LANDED], (unsigned)(nq - 1), lane); ENG_STAMP(2 + nq - 2); } }
        wait_vmcnt<0>();
        lds_post(&flags[F_LANDED], (unsigned)nq, lane);
        ENG_STAMP(2 + nq - 1);
        ENG_STAMP(15);
        return;
    }

    // -------------------------------------------------------------------- consumers
    // their x loads go out in front of everything the loader will queue on this CU (the vector-memory path returns in order)
    const u32x4 xraw = reinterpret_cast<const u32x4*>(a.x)[tid];
    u32x4 wraw = {0u, 0u, 0u, 0u};
    if (NORM) wraw = reinterpret_cast<const u32x4*>(a.rms_w)[tid];
    asm volatile("s_barrier" ::: "memory");   // nobody waits for a vector load here
    if (wave == 0) ENG_STAMP(16);
    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    float* tot = reinterpret_cast<float*>(smem + L::TOT);
    {   // x chain: the staging of gemv_q4_body (stage_tail), one chunk per thread
        float ss = 1.f;
        if (NORM) {
            part[tid] = sumsq8(xraw, 0.f);
            consumer_barrier(&flags[F_BAR0], lane, &flags[F_FAIL]);
            ss = rms_scale_from_partials<KSL * 256>(part, KSL * 256, a.K);
        }
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
        consumer_barrier(&flags[F_BAR1], lane, &flags[F_FAIL]);
    }
    if (wave == 0) ENG_STAMP(17);
    u32x4 X[KSL][4];
    float corr[KSL];
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) {
#pragma unroll
        for (int d = 0; d < 4; d++) X[ks][d] = xs[((ks * 4 + d) << 6) + lane];
        corr[ks] = sx[ks * 64 + lane];
    }
    // wave w owns (column w / 2, matrix w % 2) of every quad: its addresses inside a slot and inside the side data are constants
    const int col = wave >> 1, mat = wave & 1;
    const unsigned char* wbase = smem + L::RING + (unsigned)((mat * 4 + col) * KSL) * 1024u + lane * 16u;
    const unsigned char* sbase = smem + L::SIDE_S + mat * L::SIDE_S_BYTES + ((unsigned)col * (16u * KSL) + (lane >> 2)) * 2u;
    const unsigned char* zbase = smem + L::SIDE_Z + mat * L::SIDE_Z_BYTES + ((unsigned)col * (2u * KSL) + (lane >> 5)) * 4u;
    const unsigned zsh = ((lane >> 2) & 7u) * 4u;      // nibble of this lane's group in its zeros word (16 groups per k-slot)
    unsigned known = 0;                                // fills this wave has seen landed: the flag is read again only past it
    // the reads of one slot: this wave's two weight pieces, their scales and zero words; then the slot's release -- a slot is
    // free once every consumer's reads have been executed, and LDS executes a wave's operations in order: the bump follows them
    auto read_slot = [&](int i, u32x4 (&W)[KSL], unsigned (&zw)[KSL], uint16_t (&sc)[KSL]) {
        const unsigned char* sl = wbase + (unsigned)(i % ENG_RING) * L::SLOT;
#pragma unroll
        for (int ks = 0; ks < KSL; ks++) {             // group of unit ks * 64 + lane = ks * 16 + lane / 4
            W[ks] = *reinterpret_cast<const u32x4*>(sl + ks * 1024);
            sc[ks] = *reinterpret_cast<const uint16_t*>(sbase + (unsigned)i * (128u * KSL) + ks * 32);
            zw[ks] = *reinterpret_cast<const unsigned*>(zbase + (unsigned)i * (32u * KSL) + ks * 8);
        }
        lds_post(&flags[F_READ0 + wave], (unsigned)i + 1u, lane);   // (a plain store of this wave's own count: no atomic, no contention)
    };
    // PF (measured, off): the NEXT slot's reads go out before the current slot is multiplied whenever that slot has landed already
    // (one look at the flag, no waiting), to hide their LDS latency under ~100 VALU instructions. Slower -- 10.17 against 9.68 us
    // by rocprofv3, 940.7 against 952.2 tokens/s in one call: the register copies and the second flag read are VALU work too, and
    // a wave that runs ahead takes issue slots from the wave whose slot is due.
    u32x4 Wn[KSL];
    unsigned zn[KSL];
    uint16_t sn[KSL];
    bool have_next = false;

    for (int g4 = 0; g4 * 4 < nq; g4++) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = g4 * 4 + r;
            if (i < nq) {
                u32x4 W[KSL];
                unsigned zw[KSL];
                uint16_t sc[KSL];
                if (PF && have_next) {
#pragma unroll
                    for (int ks = 0; ks < KSL; ks++) { W[ks] = Wn[ks]; zw[ks] = zn[ks]; sc[ks] = sn[ks]; }
                } else {
                    if (known <= (unsigned)i) known = lds_wait_ge(&flags[F_LANDED], (unsigned)i + 1u, &flags[F_FAIL]);
                    read_slot(i, W, zw, sc);
                }
                if (wave == 0) ENG_STAMP(18 + i);
                have_next = false;
                if (PF && i + 1 < nq) {
                    if (known <= (unsigned)i + 1u) known = lds_peek(&flags[F_LANDED]);
                    if (known > (unsigned)i + 1u) { read_slot(i + 1, Wn, zn, sn); have_next = true; }
                }
                float c = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSL; ks++) {
                    const u32x4 w = W[ks];
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
                    const float zf = (float)((zw[ks] >> zsh) & 0xFu);
                    float t = __builtin_fmaf(acc_e, 16.f, acc_o);
                    t = __builtin_fmaf(zf, corr[ks], t);
                    c = __builtin_fmaf(h2f(sc[ks]), t, c);
                }
                cs[r] = c;
                if (STAMPS && wave == 0) { asm volatile("" : "+v"(c)); ENG_STAMP(31 + i); }
            }
        }
        const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row r: the column of slot g4 * 4 + r
        const int row = lane >> 4;
        if ((lane & 15u) == 0 && g4 * 4 + row < nq) tot[(g4 * 4 + row) * 8 + wave] = total;
    }
    consumer_barrier(&flags[F_BAR2], lane, &flags[F_FAIL]);
    if (wave == 0) ENG_STAMP(44);
    if ((int)tid < nq * 4) {
        const int i = tid >> 2, c = tid & 3;
        const float g = tot[i * 8 + c * 2], u = tot[i * 8 + c * 2 + 1];
        float val = g;
        val *= 1.0f / (1.0f + expf(-val));              // gpu_kernels.h:271
        val *= u;                                       // :272
        a.out[0][(q0 + i) * 4 + c] = lds_peek(&flags[F_FAIL]) != 0u ? (uint16_t)0x7E00u : f2h(val);   // (NaN: a wait ran out)
    }
    if (wave == 0) ENG_STAMP(45);
}
#undef ENG_STAMP

// the shapes the engine covers: K = 4096 (two 1 KiB k-slots per column), N in whole quads, at most ENG_NQMAX quads per CU
// ... on a stream that may use every CU (one 147 KiB block per CU: on a CU-masked stream the blocks would queue behind each other)
static bool ffn_engine_covers(const GemvArgs& a) {
    return g_gemv_form >= GEMV_ENGINE_LAG1 && g_gemv_form <= 7 && g_gemv_form != 4 && g_ablate == 0 && a.K == 4096 && (a.N & 3) == 0 && a.pw4 == 128 && a.sh == 32 && a.pzh == 4 &&
           divUp(a.N >> 2, cu_count()) <= ENG_NQMAX && (a.N >> 2) >= cu_count() && stream_cu_count() == cu_count();
}

template <int KSL, bool NORM, int LAG, bool STAMPS, bool PF>
static int launch_engine(const GemvArgs& a) {
    constexpr size_t smem = EngLds<KSL>::BYTES;
    { const int rc = lds_opt_in((const void*)ffn_engine_kernel<KSL, NORM, LAG, STAMPS, PF>, smem); if (rc) return rc; }
    const unsigned nquads = (unsigned)a.N >> 2, nb = (unsigned)cu_count();
    Q4_LAUNCH((ffn_engine_kernel<KSL, NORM, LAG, STAMPS, PF>), dim3(nb), dim3((ENG_CONSUMERS + 1) * 64), smem, a, nquads / nb, nquads % nb);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

template <bool NORM, bool STAMPS>
static int launch_engine_lag(const GemvArgs& a) {
    switch (g_gemv_form) {        // + 4: WITH the next-slot prefetch (profiling build; measured slower: 10.2 against 9.7 us)
        case 1: return launch_engine<2, NORM, 1, STAMPS, false>(a);
        case 2: return launch_engine<2, NORM, 2, STAMPS, false>(a);
        case 3: return launch_engine<2, NORM, 3, STAMPS, false>(a);
        case 5: return launch_engine<2, NORM, 1, STAMPS, true>(a);
        case 6: return launch_engine<2, NORM, 2, STAMPS, true>(a);
        default: return launch_engine<2, NORM, 3, STAMPS, false>(a);
    }
}

static int launch_ffn_engine(const GemvArgs& a) {
    const bool norm = a.rms_w != nullptr;
    if (a.dbg) return norm ? launch_engine_lag<true, true>(a) : launch_engine_lag<false, true>(a);
    return norm ? launch_engine_lag<true, false>(a) : launch_engine_lag<false, false>(a);
}

}  // namespace q4
