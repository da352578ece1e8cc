// attention.h -- MultiHeadAttention (llama2_q4.cu:267-284) as ONE block per head: mat_vec_kernel_simple
// (gpu_kernels.h:142-168) + softmax_kernel (:357-401) + vec_mat_kernel (:279-329) without the `att` round trip.
// A K/V row of a head is head_size halves = LPR lanes x 16 B, so one wave instruction fetches 64/LPR positions,
// coalesced. Pass 1 scores -> LDS (fp32, rounded through fp16 like the reference's `att`, :167), block max / exp /
// sum (:373-396), pass 2 probabilities (rounded through fp16, :400) times V (:311). Scores never leave the CU.
//
// The same body serves the stand-alone kernel (attention_kernel) and the attention role of the fused attention -> o-proj
// launch (layer_attn.hip, FUSED = 2), which publishes its output write-through for the o-proj blocks of the same launch.
#pragma once
#include "gemv_q4.h"
#include "lds_dma.h"

namespace q4 {

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    if (LPR >= 8) v += dpp_mov<0x141>(v);
    if (LPR >= 16) v += dpp_mov<0x140>(v);
    if (LPR >= 32) v += __shfl_xor(v, 16);
    return v;
}

constexpr int ATT_NW = 16;   // waves per block of the split-context kernels

struct AttArgs {
    q4_half* output;             // [n_heads * head_size]
    const q4_half* q;
    const q4_half* key_cache;    // already offset to the layer
    const q4_half* value_cache;
    int head_size, kv_mul, kv_dim;
    const int* pPos;
    float alpha;                 // 1/sqrt(head_size), computed in double then cast (llama2_q4.cu:273)
    int lds_scores;              // fp32 score slots in LDS (the sequence-length bin)
    unsigned long long* dbg;     // profiling stamps (profiling build)
};

// U: wave instructions in flight per pass, NW: waves per block (U * NW * R positions per pass)
// FUSED: 0 stand-alone; 2 attention role of the attention -> o-proj launch (inputs from the previous launch, output published as
// data-tagged granules for the o-proj blocks of the same launch)
// PAD: head_size is not LPR * 8 (any multiple of 8 up to 256, like the reference's kernels: heads of 80, 96, 160 ...): the lanes
// past the head's last 16-byte slice address slice 0 and multiply by a zero q
// VS > 1 (fused role only): VS = LPR / 4 blocks per head. Every one computes ALL scores of the head and the softmax (the K rows
// of all but the first of them come out of the XCD's L2: the blocks of a head share an XCD), then takes ONE 64-byte slice of
// the V rows -- 32 of the head's outputs -- with the whole wave: 4 lanes per row, 16 positions per wave instruction, so the
// P.V pass costs 1/VS of the one-block form's instructions (with two waves per SIMD that pass is an issue-latency chain:
// 1300-1950 of a head block's 6300-9600 cycles at positions 100 / 220, tools/lab/timeline_attn.py), and publishes that slice. No
// merge and no hand-off between the blocks of a head; the scores, the statistics and the rounding points are the one-block
// form's, only the fp32 order in which an output sums its positions differs (16 positions per instruction instead of 4).
// LB: positions the caller guarantees to exist whatever the position word says (bin 256 is entered at position 128): their K
// rows are requested BEFORE the word has arrived -- 0.19 us of dependent latency off half of the K stream
// VW (V-slice form): waves that take part in the P.V pass and whose partials the output sums. A sixteen-wave caller (the whole-layer launch,
// gemv_ffn_pair.h) passes 8: scores and statistics do not depend on the wave count, and with the P.V pass on eight waves every output adds the
// terms of the eight-wave role in the same order -- bit for bit the same head.
// lds_off: where the body's scratch (32 + NW * head_size + lds_scores floats) begins in the block's dynamic LDS; hook(): called once every K / V
// request of the first group is out (a caller with other streams to start puts them behind these).
struct AttNoHook { __device__ __forceinline__ void operator()() const {} };
template <int LPR, int U, int NW, int FUSED, bool PAD = false, int VS = 1, int LB = 0, int VW = NW, typename HOOK = AttNoHook>
__device__ __forceinline__ void attention_body(const AttArgs& a, const int h, const Handoff& ho, const int vs = 0, const unsigned lds_off = 0u, const HOOK hook = HOOK{}) {
    static_assert(!PAD || !FUSED, "padded heads: stand-alone kernel only");
    static_assert(VS == 1 || (FUSED && LPR == 4 * VS && (U * NW * (64 / LPR)) % (VW * 16) == 0), "V slices: fused role, 64-byte slices, whole wave instructions");
    static_assert(VW == NW || VS > 1, "fewer P.V waves: V-slice form only");
    constexpr int UV = VS > 1 ? U * NW * (64 / LPR) / (VW * 16) : U;   // V-slice form: wave instructions per pass (16 positions each)
    constexpr int R = 64 / LPR;            // positions per wave instruction
    constexpr bool PUB = FUSED != 0;
#ifdef Q4_PROFILING
    constexpr bool STAMPS = true;          // profiling build: the fused role is stamped too (tools/lab/timeline_attn.py)
#else
    constexpr bool STAMPS = !FUSED;
#endif
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // profiling stamps (dbg != nullptr only)
    if (STAMPS && a.dbg) ts[0] = __builtin_readcyclecounter();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red_max = reinterpret_cast<float*>(smem + lds_off);   // [16]
    float* red_sum = red_max + 16;                           // [16]
    float* outp = red_sum + 16;                              // [NW][head_size] output partials
    float* sc = outp + NW * a.head_size;                     // [lds_scores] scores, then exps
    const int head_size = a.head_size, kv_dim = a.kv_dim;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = tid >> 6;
    const int row = lane / LPR, sub = lane % LPR;            // position within the instruction, 16-B slice of the row
    constexpr int stride = NW * R;                           // positions per block step
    constexpr int group = stride * U;                        // positions per block pass
    const bool lane_on = !PAD || sub * 8 < head_size;
    const int subc = lane_on ? sub : 0;
    const size_t hoff = (size_t)(h / a.kv_mul) * head_size + subc * 8;   // this lane's 16-B slice inside a cache row
    const q4_half* kh = a.key_cache + hoff;
    const q4_half* vh = a.value_cache + hoff;
    if (NW < 16 && tid >= NW && tid < 16) { red_max[tid] = -INFINITY; red_sum[tid] = 0.f; }   // the reductions read 16 entries
    constexpr int ULB = LB / stride;                         // wave instructions whose rows all lie below LB
    static_assert(ULB <= U, "guaranteed rows: at most one register-resident group");
    const unsigned row_bytes = (unsigned)kv_dim * 2u;
    const unsigned lane_off = (unsigned)hoff * 2u;
    u32x4 kv0[U], vv0[UV];
    u32x4 qv;
    int pos_word;
    if constexpr (ULB > 0) {       // position word, q, then the guaranteed K rows: all in flight before the word is looked at
        pos_word = *a.pPos;
        qv = *reinterpret_cast<const u32x4*>(a.q + (size_t)h * head_size + subc * 8);
        __builtin_amdgcn_sched_barrier(0);
        const __amdgpu_buffer_rsrc_t rk_lb = __builtin_amdgcn_make_buffer_rsrc((void*)a.key_cache, 0, (unsigned)LB * row_bytes, 0x00020000);
#pragma unroll
        for (int u = 0; u < ULB; u++)
            kv0[u] = __builtin_amdgcn_raw_buffer_load_b128(rk_lb, (unsigned)(wave * R + row + u * stride) * row_bytes + lane_off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        pos_word = *a.pPos;
    }
    const int size = __builtin_amdgcn_readfirstlane(pos_word) + 1;    // wave-uniform by construction: descriptors below stay in SGPRs
    if (STAMPS && a.dbg) { asm volatile("" :: "s"(size)); ts[1] = __builtin_readcyclecounter(); }

    // ---- the first group's K AND V rows go out together: one memory latency for the whole kernel at context <= `group`
    // positions. Request order = arrival order (a wave's loads return in order), pinned with sched_barriers: q, every K row,
    // every V row -- the scores, the two softmax barriers and the exponentials run while the V rows are still landing (left
    // alone, hipcc requested q last). Buffer loads bounded at `size` rows: a row past the position is out of range and comes
    // back as zeros WITHOUT a memory request and without a branch; as `if (t < size) load` this compiled to eight exec-masked
    // blocks with an s_waitcnt vmcnt(0) in the middle of them and another one in front of the first dot product: three
    // dependent round trips, ~2 us (s_memtime stamps, tools/lab/timeline_attn.py). Rows past the position are not requested at all:
    // requesting the whole bin ahead of the position word, to save that dependent latency, was measured 22-31 us per token
    // SLOWER at 7B -- the bytes cost more.
    if constexpr (ULB == 0) qv = *reinterpret_cast<const u32x4*>(a.q + (size_t)h * head_size + subc * 8);
    if (PAD && !lane_on) qv = (u32x4){0u, 0u, 0u, 0u};
    __builtin_amdgcn_sched_barrier(0);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)a.key_cache, 0, (unsigned)size * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)a.value_cache, 0, (unsigned)size * row_bytes, 0x00020000);
#pragma unroll
    for (int u = ULB; u < U; u++)
        kv0[u] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)(wave * R + row + u * stride) * row_bytes + lane_off, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VS > 1) {      // lane = (position lane >> 2 of 16, 16-byte piece lane & 3 of the block's 64-byte slice)
        const unsigned slice_off = ((unsigned)(h / a.kv_mul) * (unsigned)head_size + (unsigned)vs * 32u + (lane & 3u) * 8u) * 2u;
        // (waves past VW aim beyond the descriptor's range: zeros, no memory request, no branch around a load)
        const bool vwave = VW == NW || wave < VW;
#pragma unroll
        for (int u = 0; u < UV; u++)
            vv0[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, vwave ? (unsigned)(wave * 16 + (int)(lane >> 2) + u * VW * 16) * row_bytes + slice_off : 0xFFFFFF00u, 0, 0);
    } else {
#pragma unroll
        for (int u = 0; u < U; u++)
            vv0[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)(wave * R + row + u * stride) * row_bytes + lane_off, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    hook();

    // ---- pass 1: scores (loop bounds are wave-uniform so DPP row sums always see full rows) ----------
    float wmax = -INFINITY;
    for (int g0 = 0; g0 < size; g0 += group) {
        u32x4 kv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (g0 == 0) {
                kv[u] = kv0[u];
            } else {
                const int t = g0 + wave * R + row + u * stride;
                const int tc = t < size ? t : size - 1;
                kv[u] = *reinterpret_cast<const u32x4*>(kh + (size_t)tc * kv_dim);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = g0 + wave * R + row + u * stride;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) s = __builtin_amdgcn_fdot2(as_h2(kv[u][e]), as_h2(qv[e]), s, false);
            s = row_sum<LPR>(s);
            s = round_h(s * a.alpha);                                             // gpu_kernels.h:164-167
            if (t < size) {
                wmax = fmaxf(wmax, s);
                if (sub == 0) sc[t] = s;
            }
        }
    }
    wmax = wave_max(wmax);
    if (STAMPS && a.dbg) ts[2] = __builtin_readcyclecounter();
    if (lane == 0) red_max[wave] = wmax;
    __syncthreads();                                                              // barrier 1: scores + wave maxima
    if (STAMPS && a.dbg) ts[3] = __builtin_readcyclecounter();

    // ---- softmax statistics (gpu_kernels.h:373-396) ------------------------------------------------
    // launches whose sequence-length bin exceeds 8192 run softmax_kernel_no_smem in the reference (llama2_q4.cu:276-279): the
    // exponential goes back to the fp16 `att` buffer (gpu_kernels.h:432) while the fp32 value is summed (:433), and the
    // normalisation divides the ROUNDED exponential (:445)
    const bool no_smem = !FUSED && a.lds_scores > 8192;
    const float m = row16_max(red_max[lane & 15]);
    float sum = 0.f;
    for (int t = tid; t < size; t += NW * 64) {
        const float e = expf(sc[t] - m);
        sc[t] = no_smem ? round_h(e) : e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red_sum[wave] = sum;
    __syncthreads();                                                              // barrier 2: exps + wave sums
    if (STAMPS && a.dbg) ts[4] = __builtin_readcyclecounter();
    // fixed order: DPP tree over the 16 wave sums
    sum = row16_sum(red_sum[lane & 15]);
    const float inv_sum = 1.0f / sum;       // one IEEE division; p = e * inv_sum is within 1 fp32 ulp of e / sum (:400)

    // ---- pass 2: att . V -----------------------------------------------------------------------------
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    if constexpr (VS > 1) {
        // (the fused role never has more than one register-resident group: size <= the bin = `group`)
#pragma unroll
        for (int u = 0; u < UV; u++) {
            const int t = wave * 16 + (int)(lane >> 2) + u * VW * 16;
            const float p = t < size && (VW == NW || wave < VW) ? round_h(sc[t] * inv_sum) : 0.f;            // gpu_kernels.h:400
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const h2 v2 = as_h2(vv0[u][e]);
                acc[2 * e] = __builtin_fmaf((float)v2.x, p, acc[2 * e]);          // :311
                acc[2 * e + 1] = __builtin_fmaf((float)v2.y, p, acc[2 * e + 1]);
            }
        }
        // the four DPP rows with the transposing swaps (row r then holds elements r and 4 + r of the lane's 16-byte piece), then
        // the four position lanes of a row that share a piece (lanes i, i + 4, i + 8, i + 12): two row rotations
        float s0 = swap16_add(swap32_add(acc[0], acc[2]), swap32_add(acc[1], acc[3]));
        float s1 = swap16_add(swap32_add(acc[4], acc[6]), swap32_add(acc[5], acc[7]));
        s0 += dpp_mov<0x128>(s0); s0 += dpp_mov<0x124>(s0);                       // row_ror:8, row_ror:4
        s1 += dpp_mov<0x128>(s1); s1 += dpp_mov<0x124>(s1);
        if ((lane & 12u) == 0u && (VW == NW || wave < VW)) {
            outp[wave * 32 + (lane & 3u) * 8 + (lane >> 4)] = s0;
            outp[wave * 32 + (lane & 3u) * 8 + 4 + (lane >> 4)] = s1;
        }
    } else {
        for (int g0 = 0; g0 < size; g0 += group) {
            u32x4 vv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (g0 == 0) {
                    vv[u] = vv0[u];
                } else {
                    const int t = g0 + wave * R + row + u * stride;
                    const int tc = t < size ? t : size - 1;
                    vv[u] = *reinterpret_cast<const u32x4*>(vh + (size_t)tc * kv_dim);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = g0 + wave * R + row + u * stride;
                const float p = t < size ? round_h(sc[t] * inv_sum) : 0.f;            // gpu_kernels.h:400
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const h2 v2 = as_h2(vv[u][e]);
                    acc[2 * e] = __builtin_fmaf((float)v2.x, p, acc[2 * e]);          // :311
                    acc[2 * e + 1] = __builtin_fmaf((float)v2.y, p, acc[2 * e + 1]);
                }
            }
        }
        // combine the R rows of a wave, then the waves through LDS. LPR == 16 (head 128): the four DPP rows are summed
        // with the transposing permlane swaps of gemv_q4.h (VALU only): afterwards row r holds the 4-row totals of
        // elements r and 4 + r of its 16-B slice -- instead of 16 ds_bpermute shuffles per lane
        if constexpr (LPR == 16 && !PAD) {
            const float s0 = swap16_add(swap32_add(acc[0], acc[2]), swap32_add(acc[1], acc[3]));   // rows: e0, e1, e2, e3
            const float s1 = swap16_add(swap32_add(acc[4], acc[6]), swap32_add(acc[5], acc[7]));   // rows: e4, e5, e6, e7
            outp[wave * head_size + sub * 8 + row] = s0;
            outp[wave * head_size + sub * 8 + 4 + row] = s1;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float v = acc[e];
                if (LPR <= 32) v += __shfl_xor(v, 32);
                if (LPR <= 16) v += __shfl_xor(v, 16);
                if (LPR <= 8) v += __shfl_xor(v, 8);
                if (LPR <= 4) v += __shfl_xor(v, 4);
                acc[e] = v;
            }
            if (lane < LPR && lane_on) {
#pragma unroll
                for (int e = 0; e < 8; e++) outp[wave * head_size + sub * 8 + e] = acc[e];
            }
        }
    }
    if (STAMPS && a.dbg) ts[5] = __builtin_readcyclecounter();
    __syncthreads();                                                              // barrier 3: output partials
    if (PUB) {
        // one granule (two outputs) per thread of wave 0, ONE store instruction per head, validated by its tag on the
        // o-proj side; the plain copy keeps RunState::xb what the launch sequence leaves there
        if ((int)tid < head_size / (2 * VS)) {
            const int g = vs * (head_size / (2 * VS)) + (int)tid;      // granule (pair of outputs) inside the head
            float s2[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int n = g * 2 + k;
                float part[VW];
#pragma unroll
                for (int w = 0; w < VW; w++) part[w] = VS > 1 ? outp[w * 32 + (int)tid * 2 + k] : outp[w * head_size + n];
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < VW; w++) s += part[w];
                s2[k] = s;
            }
            const h2 hh = {(f16_t)s2[0], (f16_t)s2[1]};
#ifdef Q4_PROFILING
            if (!ho.mute)
#endif
            store_granule(ho.pub + line_slot((unsigned)(h * (head_size / 2) + g)), as_u(hh), ho.tag);
            *reinterpret_cast<unsigned*>(a.output + (size_t)h * head_size + g * 2) = as_u(hh);
        }
        if (STAMPS && a.dbg && lane == 0 && vs == 0) {
            ts[6] = __builtin_readcyclecounter();
            unsigned long long* d = a.dbg + ((size_t)h * NW + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = ts[i];
        }
    } else {
        for (int n = tid; n < head_size; n += NW * 64) {
            float part[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) part[w] = outp[w * head_size + n];       // independent LDS reads
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) s += part[w];
            a.output[(size_t)h * head_size + n] = f2h(s);
        }
        if (a.dbg && lane == 0) {
            ts[6] = __builtin_readcyclecounter();
            unsigned long long* d = a.dbg + ((size_t)h * NW + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = ts[i];
        }
    }
}

template <int LPR, int U = 4, int NW = ATT_NW, bool PAD = false>
__global__ void __launch_bounds__(NW * 64) attention_kernel(const AttArgs a) {
    attention_body<LPR, U, NW, 0, PAD>(a, blockIdx.x, Handoff{});
}

// Long contexts: the same block, but one per (head, 256-position chunk) so that all CUs pull on the KV cache (at
// pos 2047 a layer's K+V is 32 MB; 32 single-head blocks would stream it at a few CUs' worth of bandwidth). Each
// block leaves a flash-decode partial -- un-normalised acc[head_size], running max m, sum l of exp(s - m), fp32 -- in the
// `att` scratch (RunState::att, the buffer the reference keeps its scores in); the partials are merged as
//     out = sum_s exp(m_s - M) acc_s / sum_s exp(m_s - M) l_s      (fixed order over the chunks).
// Scores are still rounded through fp16 (:167); the probabilities stay fp32 here (the reference rounds them to fp16,
// :400 -- a <= 2^-11 relative difference).
// Two ways to merge: the LAST block of a head to finish does it (arrive != nullptr: partial written through (sc1), drained,
// one returning arrival on the head's counter; nobody waits for anybody, the last arriver re-arms the counter), or
// attention_combine_kernel in a second launch (the public q4_multi_head_attention, whose caller owns the scratch).
constexpr int ATT_REC_PAD = 4;   // a partial record is head_size + 4 floats: acc[head_size], m, l, 2 x pad (16-byte rows)

// merge the nsp partial records of one head for output n. The loads of a batch of 8 chunks go out together (a loop
// of dependent loads costs one memory latency per chunk: the stand-alone combine kernel spent 4.8 us on 8 chunks that way);
// the arithmetic is strictly in chunk order, whatever the batching.
template <bool SC1>
__device__ __forceinline__ uint16_t combine_partials(q4_half* out_head, const float* base, int head_size, int nsp, int n) {
    const int rec = head_size + ATT_REC_PAD;
    auto ld = [&](const float* p) -> float {
        return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    };
    constexpr int B = 8;
    float M = -INFINITY, denom = 0.f, num = 0.f;
    if (nsp <= B) {                      // one round trip: m, l and acc[n] of every chunk
        float mv[B], lv[B], av[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const float* ps = base + (size_t)(k < nsp ? k : 0) * rec;
            mv[k] = ld(ps + head_size); lv[k] = ld(ps + head_size + 1); av[k] = ld(ps + n);
        }
#pragma unroll
        for (int k = 0; k < B; k++) if (k < nsp) M = fmaxf(M, mv[k]);
#pragma unroll
        for (int k = 0; k < B; k++) if (k < nsp) denom += lv[k] * expf(mv[k] - M);
#pragma unroll
        for (int k = 0; k < B; k++)
            if (k < nsp) { const float w = expf(mv[k] - M); num += w > 0.f ? av[k] * w : 0.f; }   // neutral chunks hold no acc
    } else {
        for (int s0 = 0; s0 < nsp; s0 += B) {
            float mv[B];
#pragma unroll
            for (int k = 0; k < B; k++) mv[k] = ld(base + (size_t)(s0 + k < nsp ? s0 + k : 0) * rec + head_size);
#pragma unroll
            for (int k = 0; k < B; k++) if (s0 + k < nsp) M = fmaxf(M, mv[k]);
        }
        for (int s0 = 0; s0 < nsp; s0 += B) {
            float mv[B], lv[B], av[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
                const float* ps = base + (size_t)(s0 + k < nsp ? s0 + k : 0) * rec;
                mv[k] = ld(ps + head_size); lv[k] = ld(ps + head_size + 1); av[k] = ld(ps + n);
            }
#pragma unroll
            for (int k = 0; k < B; k++)
                if (s0 + k < nsp) {
                    const float w = expf(mv[k] - M);
                    denom += lv[k] * w;
                    num += w > 0.f ? av[k] * w : 0.f;
                }
        }
    }
    out_head[n] = f2h(num / denom);
    return f2h(num / denom);
}

struct SplitArgs {
    float* partials;             // [heads][nsp][head_size + ATT_REC_PAD] fp32 records
    const q4_half* q;
    const q4_half* key_cache;    // already offset to the layer
    const q4_half* value_cache;
    int head_size, kv_mul, kv_dim;
    const int* pPos;
    float alpha;
    q4_half* output;
    unsigned* arrive;            // [heads] arrival counters (zero between launches), or null: a second launch merges
    unsigned long long* dbg;     // profiling build: [(head * chunks + chunk) * NW + wave][8] wall-clock stamps (tools/lab/timeline_split.py)
};

// a live chunk's flash-decode record leaves as 16-byte stores: thread n4 sums the NW wave partials of outputs 4*n4 .. 4*n4+3 (in wave
// order), thread head_size / 4 carries (m, l)
template <bool PUB, int NW>
__device__ __forceinline__ void split_store_record(const SplitArgs& a, const Handoff& ho, const float* outp, float m, float l, int h, int sp, int nsp) {
    const int head_size = a.head_size;
    const int rec = head_size + ATT_REC_PAD;
    const unsigned tid = threadIdx.x;
    if ((int)tid <= head_size / 4) {
        f32x4 r4;
        if ((int)tid < head_size / 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int n = tid * 4 + k;
                float part[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) part[w] = outp[w * head_size + n];
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) s += part[w];
                r4[k] = s;
            }
        } else {
            r4 = (f32x4){m, l, 0.f, 0.f};
        }
        if (PUB) {       // the record leaves as data-tagged granules {float, tag}: whoever merges validates every word by itself
            u32x2v* g = reinterpret_cast<u32x2v*>(a.partials) + ((size_t)h * nsp + sp) * rec + tid * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) store_granule(g + k, (unsigned)as_i(r4[k]), ho.tag);
        } else {
            float* dst = a.partials + ((size_t)h * nsp + sp) * rec + tid * 4;
            if (a.arrive) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(r4) : "memory");
            else *reinterpret_cast<f32x4*>(dst) = r4;
        }
    }
}

// PUB: the merged output also leaves as data-tagged granules for the o-proj blocks of the same launch (layer_attn.hip)
constexpr size_t att_split_lds_bytes(int nw, int head_size, int ring) { return (size_t)nw * ring * 1024 + (size_t)(32 + nw * head_size) * 4; }
// How a live chunk block takes its K / V rows in: the product reads them into registers (every row requested at entry, one memory latency for the
// chunk). exp/attention_ring.h (laboratory) brings them in on per-wave LDS-DMA rings instead: same lanes, same bytes, same sums, same speed.
struct KvInRegisters { static constexpr unsigned ring_bytes(int) { return 0u; } };

#ifdef Q4_PROFILING
// per-wave wall-clock stamps (tools/lab/timeline_split.py): [0] entry, [1] position known, [2] q landed, [3] scores done, [4] block maximum known,
// [5] P.V done, [6] record stored, [7] head merged (chunk 0)
#define SPLIT_STAMP(k) do { if (a.dbg) ts[(k)] = wall_clock64(); } while (0)
#define SPLIT_STAMP_PIN(k, v) do { if (a.dbg) { asm volatile("" : "+v"(v)); ts[(k)] = wall_clock64(); } } while (0)
#else
#define SPLIT_STAMP(k) do { } while (0)
#define SPLIT_STAMP_PIN(k, v) do { } while (0)
#endif

// one live chunk: scores, the chunk's maximum, exp, P.V, the flash-decode record (acc[head_size], m, l) -- attention_split_body's middle
template <int LPR, int U, bool PUB, int NW>
__device__ __forceinline__ void split_live(KvInRegisters, const SplitArgs& a, const Handoff& ho, const int h, const int sp, const int nsp, const int size, const int t_base,
                                           float* red_max, float* red_sum, float* outp, unsigned long long* ts) {
    constexpr int R = 64 / LPR;
    constexpr int stride = NW * R;
    const q4_half* q = a.q;
    const q4_half* key_cache = a.key_cache;
    const q4_half* value_cache = a.value_cache;
    const int head_size = a.head_size, kv_mul = a.kv_mul, kv_dim = a.kv_dim;
    const float alpha = a.alpha;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = tid >> 6;
    const int row = lane / LPR, sub = lane % LPR;
    {
        const q4_half* kh = key_cache + (size_t)(h / kv_mul) * head_size + sub * 8;
        const q4_half* vh = value_cache + (size_t)(h / kv_mul) * head_size + sub * 8;
        // non-temporal: at these contexts the KV cache (>= 0.5 GB per token) does not stay in the 256 MB Infinity Cache;
        // the one-block kernel for short contexts keeps the default policy (its 128 MB per token does: 4.1 vs 4.6 us)
        // request order = arrival order: q, every K row, every V row (see attention_body)
        const u32x4 qv = *reinterpret_cast<const u32x4*>(q + (size_t)h * head_size + sub * 8);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 kv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t_base + wave * R + row + u * stride;
            const int tc = t < size ? t : size - 1;
            kv[u] = ld_nt(reinterpret_cast<const u32x4*>(kh + (size_t)tc * kv_dim));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t_base + wave * R + row + u * stride;
            const int tc = t < size ? t : size - 1;
            vv[u] = ld_nt(reinterpret_cast<const u32x4*>(vh + (size_t)tc * kv_dim));
        }
        __builtin_amdgcn_sched_barrier(0);
        float sc[U];
        float wmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t_base + wave * R + row + u * stride;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) s = __builtin_amdgcn_fdot2(as_h2(kv[u][e]), as_h2(qv[e]), s, false);
            s = row_sum<LPR>(s);
            s = round_h(s * alpha);
            sc[u] = t < size ? s : -INFINITY;
            wmax = fmaxf(wmax, sc[u]);
        }
        wmax = wave_max(wmax);
        SPLIT_STAMP_PIN(3, wmax);
        if (lane == 0) red_max[wave] = wmax;
        __syncthreads();
        float m = row16_max(red_max[lane & 15]);
        SPLIT_STAMP_PIN(4, m);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.f;
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float p = expf(sc[u] - m);                     // 0 for masked positions (sc = -inf)
            if (sub == 0) lsum += p;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const h2 v2 = as_h2(vv[u][e]);
                acc[2 * e] = __builtin_fmaf((float)v2.x, p, acc[2 * e]);
                acc[2 * e + 1] = __builtin_fmaf((float)v2.y, p, acc[2 * e + 1]);
            }
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
        __syncthreads();
        const float l = row16_sum(red_sum[lane & 15]);
        split_store_record<PUB, NW>(a, ho, outp, m, l, h, sp, nsp);
        SPLIT_STAMP(6);
    }
}

template <int LPR, int U, bool PUB, int NW = ATT_NW, typename LOAD = KvInRegisters>
__device__ __forceinline__ void attention_split_body(const SplitArgs& a, const int h, const int sp, const int nsp, const Handoff& ho) {
    float* partials = a.partials;
    const int head_size = a.head_size;
    const int* pPos = a.pPos;
    q4_half* output = a.output;
    unsigned* arrive = a.arrive;
    constexpr int ATT_CHUNK = NW * (64 / LPR) * U;           // positions per block = one register-resident group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr unsigned RING_BYTES = LOAD::ring_bytes(NW);      // (the laboratory's LDS-DMA rings lead the block's LDS; the product has none)
#ifdef Q4_PROFILING
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SPLIT_STAMP(0);
    float* red_max = reinterpret_cast<float*>(smem + RING_BYTES);
    float* red_sum = red_max + 16;
    float* outp = red_sum + 16;                              // [NW][head_size]
    __shared__ int is_last;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = tid >> 6;
    (void)lane; (void)wave;                                  // (the stamped build writes per wave)
    const int size = __builtin_amdgcn_readfirstlane(*pPos) + 1;
    SPLIT_STAMP(1);
    const int t_base = sp * ATT_CHUNK;
    const int rec = head_size + ATT_REC_PAD;
    float* my = partials + ((size_t)h * nsp + sp) * rec;
    const bool live = t_base < size;                         // chunk entirely in the future: neutral partial (m = -inf, l = 0)
    if (NW < 16 && tid >= NW && tid < 16) { red_max[tid] = -INFINITY; red_sum[tid] = 0.f; }   // the reductions read 16 entries
    if (live) {
#ifdef Q4_PROFILING
        split_live<LPR, U, PUB, NW>(LOAD{}, a, ho, h, sp, nsp, size, t_base, red_max, red_sum, outp, ts);
#else
        split_live<LPR, U, PUB, NW>(LOAD{}, a, ho, h, sp, nsp, size, t_base, red_max, red_sum, outp, nullptr);
#endif
    } else if (PUB) {        // a chunk entirely in the future: the whole neutral record (m = -inf, l = 0, acc = 0), tagged
        if ((int)tid < rec) {
            u32x2v* g = reinterpret_cast<u32x2v*>(partials) + ((size_t)h * nsp + sp) * rec;
            store_granule(g + tid, (unsigned)as_i((int)tid == head_size ? -INFINITY : 0.f), ho.tag);
        }
    } else if (tid == 0) {
        const f32x4 r4 = {-INFINITY, 0.f, 0.f, 0.f};
        float* dst = my + head_size;
        if (arrive) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(r4) : "memory");
        else *reinterpret_cast<f32x4*>(dst) = r4;
    }
    if (PUB) {
        // ---- merge by the head's FIRST chunk block (its chunk always holds position 0): the other chunks' records arrive as
        // granules of this launch's tag, no drain, no arrival counter, no returning atomic on anybody's path (the last-arriver
        // form below cost the merging block ~2.7 us after its own chunk: write-through drain, counter round trip, dependent loads).
        // Same arithmetic and order as combine_partials.
#ifdef Q4_PROFILING
        auto dump = [&]() {
            if (a.dbg && lane == 0) {
                unsigned long long* d = a.dbg + (((size_t)h * nsp + sp) * NW + wave) * 8;
#pragma unroll
                for (int i = 0; i < 8; i++) d[i] = ts[i];
            }
        };
        if (sp != 0) dump();
#endif
        if (sp != 0 || (int)tid >= head_size / 2) return;
        const u32x2v* gh = reinterpret_cast<const u32x2v*>(partials) + (size_t)h * nsp * rec;
        constexpr int B = 8;
        float M = -INFINITY, denom = 0.f, num0 = 0.f, num1 = 0.f;
        bool ok = true;
        auto fetch = [&](int s0, float (&mv)[B], float (&lv)[B], float (&a0)[B], float (&a1)[B]) {
            for (unsigned tries = 0;; tries++) {
                bool all = true;
#pragma unroll
                for (int k = 0; k < B; k++) {
                    const unsigned kk = (unsigned)(s0 + k < nsp ? s0 + k : s0) * (unsigned)rec;
                    const u32x4 ml = load_granule2(gh, kk + (unsigned)head_size);       // {m, tag, l, tag}
                    const u32x4 aa = load_granule2(gh, kk + 2u * tid);                  // {acc[2 tid], tag, acc[2 tid + 1], tag}
                    mv[k] = as_f((int)ml[0]); lv[k] = as_f((int)ml[2]); a0[k] = as_f((int)aa[0]); a1[k] = as_f((int)aa[2]);
                    all = all && ml[1] == ho.tag && ml[3] == ho.tag && aa[1] == ho.tag && aa[3] == ho.tag;
                }
                if (all) return;
                if (tries >= POLL_LIMIT / 4 || ho.dead != 0u) { ok = false; return; }
                __builtin_amdgcn_s_sleep(4);
            }
        };
        if (nsp <= B) {                       // one set of loads
            float mv[B], lv[B], a0[B], a1[B];
            fetch(0, mv, lv, a0, a1);
#pragma unroll
            for (int k = 0; k < B; k++) if (k < nsp) M = fmaxf(M, mv[k]);
#pragma unroll
            for (int k = 0; k < B; k++) if (k < nsp) denom += lv[k] * expf(mv[k] - M);
#pragma unroll
            for (int k = 0; k < B; k++)
                if (k < nsp) { const float w = expf(mv[k] - M); num0 += w > 0.f ? a0[k] * w : 0.f; num1 += w > 0.f ? a1[k] * w : 0.f; }
        } else {
            for (int s0 = 0; s0 < nsp; s0 += B) {
                float mv[B], lv[B], a0[B], a1[B];
                fetch(s0, mv, lv, a0, a1);
#pragma unroll
                for (int k = 0; k < B; k++) if (s0 + k < nsp) M = fmaxf(M, mv[k]);
            }
            for (int s0 = 0; s0 < nsp; s0 += B) {
                float mv[B], lv[B], a0[B], a1[B];
                fetch(s0, mv, lv, a0, a1);
#pragma unroll
                for (int k = 0; k < B; k++)
                    if (s0 + k < nsp) {
                        const float w = expf(mv[k] - M);
                        denom += lv[k] * w;
                        num0 += w > 0.f ? a0[k] * w : 0.f;
                        num1 += w > 0.f ? a1[k] * w : 0.f;
                    }
            }
        }
        if (!ok && ho.dead == 0u) __hip_atomic_store(ho.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned data = (unsigned)f2h(num0 / denom) | ((unsigned)f2h(num1 / denom) << 16);
        *reinterpret_cast<unsigned*>(output + (size_t)h * head_size + 2 * tid) = data;
#ifdef Q4_PROFILING
        if (!ho.mute)
#endif
        store_granule(ho.pub + line_slot((unsigned)(h * (head_size / 2)) + tid), data, ho.tag);
#ifdef Q4_PROFILING
        SPLIT_STAMP(7);
        dump();
#endif
        return;
    }
    if (arrive == nullptr) return;
    // ---- last-arriver merge: record drained to memory, then ONE returning arrival per block ------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(arrive + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == (unsigned)nsp - 1u;
        if (is_last) __hip_atomic_store(arrive + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
    }
    __syncthreads();
    if (!is_last) return;
    {
        for (int n = tid; n < head_size; n += NW * 64)
            combine_partials<true>(output + (size_t)h * head_size, partials + (size_t)h * nsp * rec, head_size, nsp, n);
    }
}

#undef SPLIT_STAMP
#undef SPLIT_STAMP_PIN

template <int LPR, int U>
__global__ void __launch_bounds__(ATT_NW * 64) attention_split_kernel(const SplitArgs a) {
    attention_split_body<LPR, U, false>(a, blockIdx.x, blockIdx.y, gridDim.y, Handoff{});
}

static __global__ void attention_combine_kernel(q4_half* output, const float* partials, int head_size, int nsp) {
    const int h = blockIdx.x;
    for (int n = threadIdx.x; n < head_size; n += blockDim.x)
        combine_partials<false>(output + (size_t)h * head_size, partials + (size_t)h * nsp * (head_size + ATT_REC_PAD), head_size, nsp, n);
}


}  // namespace q4
