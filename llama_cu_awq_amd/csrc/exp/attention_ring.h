// exp/attention_ring.h -- LABORATORY (libllama2_q4_prof.so only; profiling knob 14, forms 7 / 8 / 9 of the attention -> o-proj launch). The
// split-context attention role with its K / V rows on LDS-DMA rings -- round 5's answer to "a chunk block pulls its 64-128 KB at the ~22 GB/s a CU's
// registers allow; the strips pull 31-32 GB/s per CU through LDS". Built, bit-identical to the register form (tests/prof_cases.py), measured LEVEL at
// every ring depth (7B, ms per token inside bin 2048, registers | rings: depth 2 1.1949 | 1.2239, 4 1.1892 | 1.1929, 6 1.1873 | 1.1911, 8 1.1886 |
// 1.2002; profiles/r05_sweep_ring_d*.txt) -- because the premise was wrong: the stamps (tools/lab/timeline_split.py, profiles/r05_timeline_split_*) show
// the launch's K / V + o-proj stream already moving at the memory system's ceiling (42 MB in 7.1 us = 5.9 TB/s at position 2042), the rings finish
// the K rows earlier and the V rows later, and the 4.9 us behind the stream are hand-off hops. Not shipped (EXPERIMENTS.md).
//
// Every wave owns a private ring of D_RING 1 KiB pieces at
// the front of the block's LDS; piece j < U is the wave's K-row instruction j (64 / LPR rows x 16 B x LPR lanes: lane l's 16 bytes land
// at byte l * 16 of the piece), piece U + j its V-row instruction j. min(D_RING, 2 U) pieces go out at entry behind q; the wave waits for
// its oldest piece with vmcnt, reads its own 16 bytes back with one ds_read_b128, re-issues the entry and multiplies -- the V pieces
// land while the scores, the block maximum and the exponentials run. The SAME lanes see the SAME bytes and add the SAME terms in the
// same order as the register form. Block LDS: [NW x D_RING KiB rings][red_max 16][red_sum 16][outp].
#pragma once
#include "../attention.h"

namespace q4 {

template <int D_RING>
struct KvOnRings { static constexpr unsigned ring_bytes(int nw) { return (unsigned)nw * D_RING * 1024u; } };

#ifdef Q4_PROFILING
#define SPLIT_STAMP(k) do { if (a.dbg) ts[(k)] = wall_clock64(); } while (0)
#define SPLIT_STAMP_PIN(k, v) do { if (a.dbg) { asm volatile("" : "+v"(v)); ts[(k)] = wall_clock64(); } } while (0)
#endif

template <int LPR, int U, bool PUB, int NW, int D_RING>
__device__ __forceinline__ void split_live(KvOnRings<D_RING>, const SplitArgs& a, const Handoff& ho, const int h, const int sp, const int nsp, const int size, const int t_base,
                                           float* red_max, float* red_sum, float* outp, unsigned long long* ts) {
    constexpr int R = 64 / LPR;
    constexpr int stride = NW * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const q4_half* q = a.q;
    const q4_half* key_cache = a.key_cache;
    const q4_half* value_cache = a.value_cache;
    const int head_size = a.head_size, kv_mul = a.kv_mul, kv_dim = a.kv_dim;
    const float alpha = a.alpha;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = tid >> 6;
    const int row = lane / LPR, sub = lane % LPR;
    {
        constexpr int T = 2 * U;                             // pieces of a wave: K-row instructions, then V-row instructions
        constexpr int D = D_RING < T ? D_RING : T;               // in flight at most
        static_assert(D <= 16, "vmcnt waits are constants up to 15");
        const unsigned row_bytes = (unsigned)kv_dim * 2u;
        const unsigned lane_off = ((unsigned)(h / kv_mul) * (unsigned)head_size + (unsigned)sub * 8u) * 2u;
        const unsigned off_last = (unsigned)(size - 1) * row_bytes + lane_off;           // rows past the position: the last row, masked below
        const unsigned off0 = (unsigned)(t_base + wave * R + row) * row_bytes + lane_off;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)key_cache, 0, (unsigned)size * row_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)value_cache, 0, (unsigned)size * row_bytes, 0x00020000);
        const unsigned wave_s = (unsigned)__builtin_amdgcn_readfirstlane(wave);            // (M0 is written from an SGPR)
        const unsigned ring = (unsigned)(uintptr_t)smem + wave_s * (D_RING * 1024u);
        auto issue = [&](int j) {                            // (j is a constant at every call site once unrolled)
            const unsigned off = off0 + (unsigned)((j < U ? j : j - U) * stride) * row_bytes;
            dma_piece(ring + (unsigned)(j % D) * 1024u, off < off_last ? off : off_last, j < U ? rk : rv, 0u);
        };
        // request order = arrival order: q, then the ring. q by an asm load: hipcc must not count it (it cannot see the pieces behind it)
        u32x4 qv;
        {
            const u32x4* pq = reinterpret_cast<const u32x4*>(q + (size_t)h * head_size + sub * 8);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(qv) : "v"(pq) : "memory");
        }
#pragma unroll
        for (int j = 0; j < D; j++) issue(j);
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(qv) : "n"(D) : "memory");               // all but the ring: q has landed
        SPLIT_STAMP(2);
        const unsigned char* rbase = smem + wave_s * (D_RING * 1024u) + lane * 16u;
        float sc[U];
        float wmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < U; u++) {
            constexpr int DM1 = D - 1;
            wait_vmcnt_upto15(T - 1 - u < DM1 ? T - 1 - u : DM1);                        // piece u has landed
            const u32x4 kv = *reinterpret_cast<const u32x4*>(rbase + (u % D) * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            // read: the entry may be refilled
            if (u + D < T) issue(u + D);
            const int t = t_base + wave * R + row + u * stride;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) s = __builtin_amdgcn_fdot2(as_h2(kv[e]), as_h2(qv[e]), s, false);
            s = row_sum<LPR>(s);
            s = round_h(s * alpha);
            sc[u] = t < size ? s : -INFINITY;
            wmax = fmaxf(wmax, sc[u]);
            asm volatile("" : "+v"(wmax), "+v"(sc[u]));      // the piece's arithmetic stays in front of the next piece's wait
        }
        wmax = wave_max(wmax);
        SPLIT_STAMP_PIN(3, wmax);
        if (lane == 0) red_max[wave] = wmax;
        block_barrier_lds();
        float m = row16_max(red_max[lane & 15]);
        SPLIT_STAMP_PIN(4, m);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.f;
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++) {
            constexpr int DM1 = D - 1;
            const float p = expf(sc[u] - m);                 // 0 for masked positions (sc = -inf)
            if (sub == 0) lsum += p;
            wait_vmcnt_upto15(U - 1 - u < DM1 ? U - 1 - u : DM1);                        // piece U + u has landed
            const u32x4 vv = *reinterpret_cast<const u32x4*>(rbase + ((U + u) % D) * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (U + u + D < T) issue(U + u + D);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const h2 v2 = as_h2(vv[e]);
                acc[2 * e] = __builtin_fmaf((float)v2.x, p, acc[2 * e]);
                acc[2 * e + 1] = __builtin_fmaf((float)v2.y, p, acc[2 * e + 1]);
            }
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
        }
        SPLIT_STAMP_PIN(5, acc[0]);
        lsum = wave_sum(lsum);
        if (lane == 0) red_sum[wave] = lsum;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float v = acc[e];
            if (LPR <= 32) v += __shfl_xor(v, 32);
            if (LPR <= 16) v += __shfl_xor(v, 16);
            if (LPR <= 8) v += __shfl_xor(v, 8);
            if (LPR <= 4) v += __shfl_xor(v, 4);
            acc[e] = v;
        }
        if (lane < LPR) {
#pragma unroll
            for (int e = 0; e < 8; e++) outp[wave * head_size + sub * 8 + e] = acc[e];
        }
        block_barrier_lds();
        const float l = row16_sum(red_sum[lane & 15]);
        split_store_record<PUB, NW>(a, ho, outp, m, l, h, sp, nsp);
        SPLIT_STAMP(6);
    }
}
#undef SPLIT_STAMP
#undef SPLIT_STAMP_PIN

}  // namespace q4
