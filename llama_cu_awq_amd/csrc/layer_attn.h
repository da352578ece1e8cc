// layer_attn.h -- MultiHeadAttention (llama2_q4.cu:320) and the output projection with residual add (:323) as ONE
// launch (fusion level 3, the default).
//
// Every launch of the decode step carries ~3-4 us that are not streaming (boundary, kernel arguments, first-data latency,
// the x chain); attention and o-proj are the two latency-bound launches of a layer, so they share one: blocks [0, natt) are
// the attention role -- head_size / 32 blocks per head up to bin 256 (each: all scores of the head, one 64-byte slice of the V
// rows; attention.h, VS), or one per (head, chunk) in the split-context bins; the stand-alone kernels' bodies in 8-wave blocks
// --, the rest are the o-proj role (gemv_q4_body, 8 waves), which puts its weights in flight at entry, polls ONE granule of
// the attention output, then reads the vector (every 8-byte {two halves, tag} granule validates itself against the launch's
// epoch) and runs the dot products. The attention blocks publish with one store instruction each; below bin 512 they never wait
// (in the split-context bins a head's first chunk block gathers the other chunks' records, which wait for nobody).
//
// What the protocol rests on:
//  * forward progress: consumers wait only for producers, producers wait for nobody, and the launcher admits the form only
//    when the consumers alone cannot fill the stream's CUs (occupancy x CUs > o-proj blocks, attention_oproj_form): a slot
//    is then always open to a producer; no dispatch order is assumed. Otherwise the layer runs the stand-alone launches.
//  * the tag is the value of the model's epoch word at entry. The word is advanced by the PRECEDING launch of the stream
//    (the fused QKV GEMV, gemv_q4.h `bump`), never by this one: all blocks read the same value, whatever their timing.
//  * every wait is a bounded poll; one that runs out sets the model's sticky error word, which the token loops turn into a
//    clean retry at fusion level 1 (q4_runtime.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "attention.h"

namespace q4 {

// Columns per wave of the o-proj role: dim / (8 * columns) blocks. K = dim in two k-slots (dim <= 4096): TWO -- 256 blocks of 16
// columns at 7B, every CU of the device takes part in the dot products (with four columns per wave 128 CUs did them while the
// CUs of the finished attention blocks idled): 7B -n 256 958.7 -> 972.2, -n 2048 858.6 -> 864.5 tokens/s (tools/ab.py, one call).
// Longer K: four -- at 13B two columns per wave are 320 + 160 blocks, two per CU on most CUs: 534.4 -> 527.2.
constexpr int la_ocols(int slots) { return slots <= 2 ? 2 : 4; }
constexpr int LA_WAVES = 8;          // 512-thread blocks for both roles (16-wave o-proj blocks -- 64 of them -- lost 13-20 us per token)

struct AttOprojArgs {
    GemvArgs oproj;
    AttArgs att;
    SplitArgs split;
    unsigned* sync;                  // hand-off words of the model (sync layout: q4_internal.h)
    unsigned nheads, natt, no;       // heads, attention blocks (heads x V slices, or heads x chunks), o-proj blocks
    unsigned hold_q16;               // split-context forms: the o-proj role requests its weights hold_q16 / 65536 x 10 ns per context position after entry (0: at entry)
    unsigned long long* dbg;         // profiling build: [block][4] wall-clock stamps (entry, after the wait, end, role)
    int mute;                        // profiling build: the attention blocks do not publish (a real time-out for tests/prof_cases.py)
};

// ATT 0 / 1: one block per head, 128 / 256 positions per register-resident pass; 2 / 3 / 4: one block per (head, 128 / 256 / 64
// positions), merged by each head's first chunk block; 5 / 6: 0 / 1 with head_size / 32 blocks per head, each taking one 64-byte
// slice of the V rows (attention.h, VS; the default below bin 512). LPR = lanes per cache row of a head (head_size / 8).
// (Round 5's K / V rows on LDS-DMA rings -- forms 7 / 8 / 9 -- measured level and left the tree in round 6: EXPERIMENTS.md #29.)
constexpr size_t AO_LDS_MAX = 64 * 1024;     // block LDS of the launch: the default 64 KiB, no opt-in
constexpr bool att_is_split(int att) { return att >= 2 && att <= 4; }
template <int LPR, int ATT>
struct AttShape {
    static constexpr int CHUNK = ATT == 4 ? 64 : ATT == 0 || ATT == 2 || ATT == 5 ? 128 : 256;
    static constexpr int U = CHUNK / (LA_WAVES * (64 / LPR));
    static constexpr int VS = ATT == 5 || ATT == 6 ? LPR / 4 : 1;
    static constexpr int LB = ATT == 1 || ATT == 6 ? 128 : 0;   // forms entered above position 127 only (bins > 128)
};

// The eight pointers every block needs FIRST lead the argument list: built with kernel-argument preload (csrc/Makefile, ATTNFLAGS) they arrive in SGPRs with the
// wave -- hand-off words (epoch), position word, q, K and V rows for the attention role; the o-proj role's weight, zero and scale tensors -- instead of behind
// a scalar round trip to the argument segment; the struct carries everything else (and the same eight values, unused).
template <int SLOTS, bool HALF, int ATT, int LPR>
__global__ void __launch_bounds__(LA_WAVES * 64) attention_oproj_kernel(unsigned* const lead_sync, const int* const lead_pos, const q4_half* const lead_q, const q4_half* const lead_k,
                                                                        const q4_half* const lead_v, const uint32_t* const lead_w, const uint32_t* const lead_z, const q4_half* const lead_s,
                                                                        const AttOprojArgs a0) {
    constexpr int NW = LA_WAVES;
    AttOprojArgs a = a0;
    if constexpr (att_is_split(ATT)) {        // the split-context forms only: below bin 512 the role's entry code is hand-placed and lost 1.3 % with these copies
        a.sync = lead_sync;
        a.split.pPos = lead_pos; a.split.q = lead_q; a.split.key_cache = lead_k; a.split.value_cache = lead_v;
        a.oproj.m[0].w = lead_w; a.oproj.m[0].z = lead_z; a.oproj.m[0].s = lead_s;
    }
    constexpr int U = AttShape<LPR, ATT>::U;
    static_assert(U >= 1, "rows in flight");
    const unsigned b = blockIdx.x;
    unsigned* const sync_words = a.sync;
    Handoff ho = {};
    ho.error = sync_words + SYNC_ERROR;
    // one 8-byte sc1 load: [0] the error word, [1] the epoch. Through a lane-held (opaque) offset on purpose: for a load the
    // compiler can prove wave-uniform it puts s_waitcnt vmcnt(0) + v_readfirstlane right here, and every block of the launch then
    // starts with a second dependent memory round trip (kernel arguments -> this word) before it requests anything else
    // (seen in the ISA; ~0.5 us of the launch). Like this the wait sits at the first use: the polls, or the publishing store.
    unsigned zero_off = 0;
    asm volatile("" : "+v"(zero_off));
    const u32x2v ee = load_granule(reinterpret_cast<const u32x2v*>(sync_words), zero_off);
    ho.tag = ee[1];          // (the fused QKV launch in front has advanced it: >= 1)
    ho.dead = ee[0];
#ifdef Q4_PROFILING
    ho.mute = a.mute != 0;
#endif
    u32x2v* g_att = reinterpret_cast<u32x2v*>(sync_words + SYNC_GRANULES);
#ifdef Q4_PROFILING
    if (a.dbg && threadIdx.x == 0) { a.dbg[b * 4 + 0] = wall_clock64(); a.dbg[b * 4 + 3] = b < a.natt ? 1 : 2; }
    ho.stamp = a.dbg ? a.dbg + b * 4 + 1 : nullptr;
#endif
    if (b < a.natt) {
        ho.pub = g_att;
        if constexpr (ATT <= 1) attention_body<LPR, U, NW, 2, false, 1, AttShape<LPR, ATT>::LB>(a.att, (int)b, ho);
        else if constexpr (ATT == 5 || ATT == 6) attention_body<LPR, U, NW, 2, false, AttShape<LPR, ATT>::VS, AttShape<LPR, ATT>::LB>(a.att, (int)(b % a.nheads), ho, (int)(b / a.nheads));
        else attention_split_body<LPR, U, true, NW>(a.split, (int)(b % a.nheads), (int)(b / a.nheads), (int)(a.natt / a.nheads), ho);
    } else {
        const unsigned j = b - a.natt;
        if constexpr (att_is_split(ATT)) {
            // The chunk blocks' K / V rows (16 KB per position at 7B: 33.5 MB at 2048) move at the memory system's ceiling (5.6-5.9 TB/s,
            // tools/lab/timeline_split.py); this role's weights (8.7 MB) are not needed before the heads are merged, 4-5 us after that stream
            // ends. Requested at entry they are a fifth of the stream's bytes, in front of rows the whole launch waits for; so they are
            // requested when the stream is about to end: a time per context position priced by the host from the rows' bytes.
            if (a.hold_q16 != 0u) {
                const unsigned long long t_entry = wall_clock64();
                const unsigned size = (unsigned)(*lead_pos + 1);
                ho.hold_until = t_entry + (((unsigned long long)size * a.hold_q16) >> 16);
            }
        }
        // (Below the split-context bins the role requests at entry. Held back by a fixed 0.3 / 0.6 / 1.0 / 1.5 / 2.0 us behind the heads' K / V requests, round 6:
        //  1022.3 / 1022.2 / 1022.2 / 1015.8 / 1004.6 against 1021.4 tokens/s -- level, then worse. EXPERIMENTS.md #45)
        ho.sub = g_att;
        ho.sentinel = (int)((j % a.nheads) * (a.att.head_size / 2) + a.att.head_size / 2 - 1);   // last granule of head j % heads
        gemv_q4_body<MODE_PLAIN, SLOTS, la_ocols(SLOTS), false, 5, 1, HALF, ROLE_CONSUMER>(a.oproj, j, 0, ho);
    }
#ifdef Q4_PROFILING
    if (a.dbg && threadIdx.x == 0) a.dbg[b * 4 + 2] = wall_clock64();
#endif
}

// The same launch on SIXTEEN-wave blocks, for Llama-2-7B's shape below the split-context bins (128-wide heads, K = dim = 4096 in two k-slots): the attention
// role's (head, V slice) units take half as many wave instructions per wave for the scores (attention_body on sixteen waves, the P.V pass on eight: the
// eight-wave role's bits, attention.h VW) -- round 6's whole-layer experiment measured the role at 2.7-3.0 us against 3.2-4.1 on eight waves --, and the o-proj
// role multiplies ONE column per wave (dim / 16 blocks): reduce4_q4(c, 0, 0, 0) is row 0 of the two-column form's reduction, bit for bit. Same protocol.
// ATT 5 / 6 as above (bins 128 / 256).
constexpr int LA16_WAVES = 16;
constexpr size_t LA16_LDS = 16 * 1024;     // the attention role's scratch (32 + 16 x 128 + 256 floats) or the o-proj role's staged vector (8.5 / 12.75 KiB)
// K5120: K = dim = 5120 (Llama-2-13B): two whole k-slots and the shared half slot of gemv_q4.h -- an even column's half-slot terms in lanes 0 .. 31, an odd
// column's in lanes 32 .. 63, added (not fused) like there, so the reduction sees what the four-column form's sees
template <int ATT, bool K5120 = false>
__global__ void __launch_bounds__(LA16_WAVES * 64) attention_oproj16_kernel(unsigned* const lead_sync, const int* const lead_pos, const q4_half* const lead_q, const q4_half* const lead_k,
                                                                          const q4_half* const lead_v, const uint32_t* const lead_w, const uint32_t* const lead_z, const q4_half* const lead_s,
                                                                          const AttOprojArgs a0) {
    static_assert(ATT == 5 || ATT == 6, "V-slice forms");
    // (the eight pointers a block needs first lead the argument list: preloaded into SGPRs with the wave, see attention_oproj_kernel)
    AttOprojArgs a = a0;
    a.sync = lead_sync;
    a.att.pPos = lead_pos; a.att.q = lead_q; a.att.key_cache = lead_k; a.att.value_cache = lead_v;
    a.oproj.m[0].w = lead_w; a.oproj.m[0].z = lead_z; a.oproj.m[0].s = lead_s;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* const sync_words = a.sync;
    unsigned zero_off = 0;
    asm volatile("" : "+v"(zero_off));
    const u32x2v ee = load_granule(reinterpret_cast<const u32x2v*>(sync_words), zero_off);     // [0] the error word, [1] the epoch (see attention_oproj_kernel)
    u32x2v* const g_att = reinterpret_cast<u32x2v*>(sync_words + SYNC_GRANULES);
    if (b < a.natt) {
        Handoff ho = {};
        ho.error = sync_words + SYNC_ERROR;
        ho.tag = ee[1];
        ho.dead = ee[0];
        ho.pub = g_att;
#ifdef Q4_PROFILING
        ho.mute = a.mute != 0;
#endif
        attention_body<16, ATT == 5 ? 2 : 4, LA16_WAVES, 2, false, 4, ATT == 6 ? 128 : 0, 8>(a.att, (int)(b % a.nheads), ho, (int)(b / a.nheads));
        return;
    }
    // ---- o-proj role: column n = 16 j + wave, its two k-slots requested at entry (gemv_q4.h's loads), then ONE granule polled by one lane, the vector gathered
    // by waves 0 .. 7 (every granule validates itself), staged as gemv_q4_body stages a vector that is multiplied as it is
    const unsigned j = b - a.natt;
    const GemvArgs& o = a.oproj;
    const unsigned n = 16u * j + (unsigned)wave;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)o.m[0].w, 0, o.N * o.pw4 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)o.m[0].z, 0, o.N * o.pzh * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o.m[0].s, 0, o.N * o.sh * 2, 0x00020000);
    constexpr int NS = K5120 ? 3 : 2;                // k-slots (the last one of K = 5120 is the half slot: units 128 .. 159)
    u32x4 W[NS];
    unsigned ZW[NS];
    uint16_t SC[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const unsigned jj = K5120 && s == 2 ? 128u + (lane & 31u) : 64u * (unsigned)s + lane;
        ZW[s] = __builtin_amdgcn_raw_buffer_load_b32(rz, (jj >> 5) * 4u, n * (unsigned)o.pzh * 4u, Q4_ZS_AUX);
        SC[s] = __builtin_amdgcn_raw_buffer_load_b16(rs, (jj >> 2) * 2u, n * (unsigned)o.sh * 2u, Q4_ZS_AUX);
        W[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, jj * 16u, n * (unsigned)o.pw4 * 16u, Q4_W_AUX);
    }
    // the residual too (nobody else writes x in this launch): read at the end it is one more dependent round trip in the launch's tail
    uint16_t resid = 0;
    if (o.accum && (int)n < o.N) resid = o.out[0][n];
    __builtin_amdgcn_sched_barrier(0);
    const unsigned tag = ee[1], dead = ee[0];
    unsigned* const error = sync_words + SYNC_ERROR;
    if (tid == 0 && dead == 0u) {
        const unsigned sentinel = (j % a.nheads) * ((unsigned)a.att.head_size >> 1) + ((unsigned)a.att.head_size >> 1) - 1u;
        unsigned i = 0;
        while (load_granule(g_att, line_slot(sentinel))[1] != tag && ++i < POLL_LIMIT) __builtin_amdgcn_s_sleep(2);
        if (i >= POLL_LIMIT) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                    // [NS][4][64] x 16 B permuted x
    float* sx = reinterpret_cast<float*>(smem + NS * 4096);       // [NS][64] -(sum of the 32 x) * 2^-20
    if (tid < (K5120 ? 640u : 512u)) {                            // one 8-half chunk per thread: K = 4096 / 5120
        u32x4 xraw = {0u, 0u, 0u, 0u};
        bool ok = true;
        for (unsigned tries = 0;; tries++) {
            const u32x4 g01 = load_granule2(g_att, line_slot(tid * 4u)), g23 = load_granule2(g_att, line_slot(tid * 4u + 2u));
            xraw = (u32x4){g01[0], g01[2], g23[0], g23[2]};
            if (g01[1] == tag && g01[3] == tag && g23[1] == tag && g23[3] == tag) break;
            if (tries >= POLL_LIMIT / 4 || dead != 0u) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && dead == 0u) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned sgn = q4_stage_sign_bits(tid);
        const u32x4 pv = permute_x8(q4_signed_x(xraw, sgn));
        const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
        float cb = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
        cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);
        const unsigned u = tid >> 2, d = tid & 3u;
        xs[(((u >> 6) * 4 + d) << 6) + (u & 63u)] = pv;
        if (d == 0) sx[u] = cb * -9.5367431640625e-07f;
    }
    __syncthreads();
    float c = 0.f;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const bool hs = K5120 && s == 2;
        const unsigned lu = hs ? (lane & 31u) : lane;
        u32x4 X[4];
#pragma unroll
        for (int d = 0; d < 4; d++) X[d] = xs[((s * 4 + d) << 6) + lu];
        const float corr = sx[s * 64 + lu];
        float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const unsigned ww = W[s][d], tt = ww >> 8;
            acc_e = __builtin_amdgcn_fdot2(as_h2(ww & 0x000F000Fu), as_h2(X[d][0]), acc_e, false);
            acc_o = __builtin_amdgcn_fdot2(as_h2(ww & 0x00F000F0u), as_h2(X[d][1]), acc_o, false);
            acc_e = __builtin_amdgcn_fdot2(as_h2(tt & 0x000F000Fu), as_h2(X[d][2]), acc_e, false);
            acc_o = __builtin_amdgcn_fdot2(as_h2(tt & 0x00F000F0u), as_h2(X[d][3]), acc_o, false);
        }
        const float zf = (float)((ZW[s] >> (((lu >> 2) & 7u) * 4u)) & 0xFu);
        float t = __builtin_fmaf(acc_e, 16.f, acc_o);
        t = __builtin_fmaf(zf, corr, t);
        if (hs) {                   // gemv_q4.h's shared half slot: lanes 0 .. 31 work for the pair's even column, lanes 32 .. 63 for the odd one
            const float v = h2f(SC[s]) * t;
            const bool mine = ((lane >> 5) & 1u) == (n & 1u);
            c += mine ? v : 0.f;
        } else {
            c = __builtin_fmaf(h2f(SC[s]), t, c);
        }
    }
    const float tot = reduce4_q4(c, 0.f, 0.f, 0.f) * 1048576.f;
    if (lane == 0 && (int)n < o.N) {
        q4_half* out = o.out[0];
        float r = tot;
        if (o.accum) r += h2f(resid);                             // gpu_kernels.h:229-230
        out[n] = f2h(r);                                          // :231
    }
}
int launch_attention_oproj16(int att, dim3 grid, const AttOprojArgs& a, int* max_blocks_per_cu);   // layer_attn.hip

// one translation unit per head size (layer_attn_h64.hip, layer_attn.hip, layer_attn_h256.hip)
typedef int (*AttOprojLaunch)(int slots_kind, int att, dim3 grid, dim3 block, size_t smem, const AttOprojArgs& a, int* max_blocks_per_cu);
int launch_attention_oproj_h64(int slots_kind, int att, dim3 grid, dim3 block, size_t smem, const AttOprojArgs& a, int* max_blocks_per_cu);
int launch_attention_oproj_h128(int slots_kind, int att, dim3 grid, dim3 block, size_t smem, const AttOprojArgs& a, int* max_blocks_per_cu);
int launch_attention_oproj_h256(int slots_kind, int att, dim3 grid, dim3 block, size_t smem, const AttOprojArgs& a, int* max_blocks_per_cu);

// slots_kind: 0 = K in two k-slots, 1 = three with a shared half slot, 2 = four. With max_blocks_per_cu != nullptr the call
// only reports how many blocks of this instantiation fit a CU (occupancy query) and launches nothing.
#define Q4_AO_DISPATCH(LPR)                                                                                              \
    {                                                                                                                     \
        auto go = [&](auto kernel) -> int {                                                                               \
            { const int rc = lds_opt_in((const void*)kernel, smem); if (rc) return rc; }                                  \
            if (max_blocks_per_cu) {                                                                                      \
                int n = 0;                                                                                                \
                Q4_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, (int)block.x, smem));                     \
                *max_blocks_per_cu = n;                                                                                   \
                return Q4_OK;                                                                                             \
            }                                                                                                             \
            Q4_LAUNCH(kernel, grid, block, smem, a.sync, a.att.pPos, a.att.q, a.att.key_cache, a.att.value_cache, a.oproj.m[0].w, a.oproj.m[0].z, a.oproj.m[0].s, a); \
            Q4_LAUNCH_CHECK();                                                                                            \
            return Q4_OK;                                                                                                 \
        };                                                                                                                \
        switch (slots_kind * 16 + att) {                                                                                  \
            case 0: return go(attention_oproj_kernel<2, false, 0, LPR>);                                                  \
            case 1: return go(attention_oproj_kernel<2, false, 1, LPR>);                                                  \
            case 2: return go(attention_oproj_kernel<2, false, 2, LPR>);                                                  \
            case 3: return go(attention_oproj_kernel<2, false, 3, LPR>);                                                  \
            case 4: return go(attention_oproj_kernel<2, false, 4, LPR>);                                                  \
            case 5: return go(attention_oproj_kernel<2, false, 5, LPR>);                                                  \
            case 6: return go(attention_oproj_kernel<2, false, 6, LPR>);                                                  \
            case 16: return go(attention_oproj_kernel<3, true, 0, LPR>);                                                  \
            case 17: return go(attention_oproj_kernel<3, true, 1, LPR>);                                                  \
            case 18: return go(attention_oproj_kernel<3, true, 2, LPR>);                                                  \
            case 19: return go(attention_oproj_kernel<3, true, 3, LPR>);                                                  \
            case 20: return go(attention_oproj_kernel<3, true, 4, LPR>);                                                  \
            case 21: return go(attention_oproj_kernel<3, true, 5, LPR>);                                                  \
            case 22: return go(attention_oproj_kernel<3, true, 6, LPR>);                                                  \
            case 32: return go(attention_oproj_kernel<4, false, 0, LPR>);                                                 \
            case 33: return go(attention_oproj_kernel<4, false, 1, LPR>);                                                 \
            case 34: return go(attention_oproj_kernel<4, false, 2, LPR>);                                                 \
            case 35: return go(attention_oproj_kernel<4, false, 3, LPR>);                                                 \
            case 36: return go(attention_oproj_kernel<4, false, 4, LPR>);                                                 \
            case 37: return go(attention_oproj_kernel<4, false, 5, LPR>);                                                 \
            case 38: return go(attention_oproj_kernel<4, false, 6, LPR>);                                                 \
        }                                                                                                                 \
        return Q4_ERR_UNSUPPORTED_SIZE;                                                                                   \
    }

}  // namespace q4
