// q4_sampling.hip -- temperature / top-p sampling: sample() sampler.h:51-81 with softmax_logits_kernel
// (gpu_kernels.h:499-550) and sample_top_p_kernel (:555-584). The reference leans on cub::DeviceRadixSort and
// cub::DeviceScan (7+ launches); here ONE 1024-thread block does the whole step with the same fp16 rounding points:
//   * softmax: logits/T, exp, normalise rounded to fp16 exactly where the reference stores them; the vocabulary
//     lives in registers (32 strided elements per thread) between the phases; sums are a balanced pairwise tree over
//     the 1024 per-thread partials (cub's order is unspecified; the oracle restates this one)
//   * sort: stable LSD radix sort of the fp16 probabilities (non-negative -> the bit pattern is monotonic), two passes of
//     8 bits, descending, ties keep ascending index (what cub's stable SortPairsDescending yields). Each wave owns a
//     contiguous segment and ranks 64 keys at a time with ballot matching, so loads are coalesced and the order is
//     deterministic (no atomics)
//   * inclusive prefix sum IN FP16 (sampler.h:72-78) in a fixed cub-shaped order -- thread-sequential over 1/1024th of
//     the vocabulary, Hillis-Steele across the 64 lanes, sequential across the 16 waves, every add rounded to fp16 --
//     and the first index whose prefix reaches the threshold (:566-577). cub's order is unspecified; the oracle
//     restates this one.
#include <hip/hip_runtime.h>
#include <string.h>
#include "q4_device.h"
#include "q4_internal.h"
using namespace q4;

namespace {

// profiling build: wall-clock stamps (100 MHz) of the kernel's phases, written by thread 0 into the sampler's `indices` scratch
// (which nothing else uses) as 64-bit words: tools/lab/sampler_kernel_time.py prints them
#ifdef Q4_PROFILING
#define SMP_STAMP(k) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(indices)[k] = wall_clock64(); } while (0)
#else
#define SMP_STAMP(k) do { } while (0)
#endif

constexpr int SMP_T = 1024, SMP_W = 16, SMP_E = 32;   // threads, waves, register-resident elements per thread

__device__ __forceinline__ float block_tree_sum(float v, float* red) {   // red: 16 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return row16_sum(red[lane & 15]);
}
__device__ __forceinline__ float block_tree_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return row16_max(red[lane & 15]);
}

// softmax_logits_kernel gpu_kernels.h:499-550, in place on `logits`, indices[t] = t (:507)
// Returns (register path) the largest probability and its lowest index packed as prob_bits << 16 | (0xFFFF - index)
// -- the first entry of the descending stable sort -- or 0 (memory path).
__device__ unsigned softmax_phase(q4_half* __restrict__ logits, int size, float temperature, int* __restrict__ indices, float* red) {
    const int tid = threadIdx.x;
    unsigned top = 0;
    if (size <= SMP_T * SMP_E) {
        float v[SMP_E];
#pragma unroll
        for (int k = 0; k < SMP_E; k++) {
            const int t = tid + k * SMP_T;
            float val = t < size ? h2f(logits[t]) : 0.f;
            val /= temperature;                                    // :511
            v[k] = round_h(val);
        }
        float max_val = -3.402823466e+38f;                         // :522
#pragma unroll
        for (int k = 0; k < SMP_E; k++)
            if (tid + k * SMP_T < size) max_val = fmaxf(max_val, v[k]);
        SMP_STAMP(1);
        max_val = block_tree_max(max_val, red);
        SMP_STAMP(2);
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < SMP_E; k++)
            if (tid + k * SMP_T < size) {
                const float e = expf(v[k] - max_val);              // :536
                v[k] = round_h(e);
                sum += e;
            }
        SMP_STAMP(3);
        sum = block_tree_sum(sum, red);
        SMP_STAMP(4);
#pragma unroll
        for (int k = 0; k < SMP_E; k++) {
            const int t = tid + k * SMP_T;
            if (t < size) {
                const uint16_t pb = f2h(v[k] / sum);                                 // :549
                logits[t] = pb;                                                       // (indices[t] = t, :507, is scratch nobody reads:
                const unsigned key = ((unsigned)pb << 16) | (0xFFFFu - (unsigned)t);   //  the sort below carries the index inside its keys)
                top = key > top ? key : top;
            }
        }
        // block maximum of the packed keys (probabilities are non-negative: the bit pattern orders them)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const unsigned o = __shfl_xor(top, off); top = o > top ? o : top; }
        unsigned* redu = reinterpret_cast<unsigned*>(red);
        __syncthreads();
        if ((tid & 63) == 0) redu[tid >> 6] = top;
        __syncthreads();
        top = redu[0];
#pragma unroll
        for (int w = 1; w < SMP_W; w++) top = redu[w] > top ? redu[w] : top;
    } else {   // larger vocabularies: the same arithmetic through memory
        for (int t = tid; t < size; t += SMP_T) {
            indices[t] = t;
            float val = h2f(logits[t]);
            val /= temperature;
            logits[t] = f2h(val);
        }
        __syncthreads();
        float max_val = -3.402823466e+38f;
        for (int i = tid; i < size; i += SMP_T) max_val = fmaxf(max_val, h2f(logits[i]));
        max_val = block_tree_max(max_val, red);
        float sum = 0.0f;
        for (int i = tid; i < size; i += SMP_T) {
            const float e = expf(h2f(logits[i]) - max_val);
            logits[i] = f2h(e);
            sum += e;
        }
        sum = block_tree_sum(sum, red);
        for (int t = tid; t < size; t += SMP_T) logits[t] = f2h(h2f(logits[t]) / sum);
    }
    __syncthreads();
    return top;
}

// lanes of this wave (among `valid` ones) holding the same 8-bit digit
__device__ __forceinline__ unsigned long long match8(unsigned d, bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// one stable pass, 8-bit digit at `shift`, descending (digit' = 255 - digit), through global memory (vocabularies
// above 32 x 1024 entries). vin == nullptr: value = index. cnt: [256][16] counters (digit-major, wave-minor), wtot: [16]
__device__ void radix_pass(const uint16_t* __restrict__ kin, const int* __restrict__ vin, uint16_t* __restrict__ kout,
                           int* __restrict__ vout, int n, int shift, volatile unsigned* cnt, unsigned* wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = ((n + SMP_T - 1) / SMP_T) * 64;              // keys per wave segment
    const int seg0 = wave * S;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i = tid; i < 256 * SMP_W; i += SMP_T) cnt[i] = 0;
    __syncthreads();
    for (int it = 0; it < S; it += 64) {
        const int i = seg0 + it + lane;
        const bool valid = i < n;
        const unsigned k = valid ? kin[i] : 0u;
        const unsigned d = 255u - ((k >> shift) & 255u);
        const unsigned long long m = match8(d, valid);
        if (valid && (m & lt) == 0ull) cnt[d * SMP_W + wave] += (unsigned)__popcll(m);   // leader of its digit
    }
    __syncthreads();
    unsigned a[4], tot = 0;                                    // exclusive scan, thread t owns counters 4t .. 4t+3
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = cnt[tid * 4 + e]; tot += a[e]; }
    unsigned incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned run = incl - tot;
    for (int w = 0; w < wave; w++) run += wtot[w];
#pragma unroll
    for (int e = 0; e < 4; e++) { cnt[tid * 4 + e] = run; run += a[e]; }
    __syncthreads();
    for (int it = 0; it < S; it += 64) {
        const int i = seg0 + it + lane;
        const bool valid = i < n;
        const unsigned k = valid ? kin[i] : 0u;
        const int v = valid ? (vin ? vin[i] : i) : 0;
        const unsigned d = 255u - ((k >> shift) & 255u);
        const unsigned long long m = match8(d, valid);
        const unsigned rank = (unsigned)__popcll(m & lt);
        if (valid) {
            const unsigned pos = cnt[d * SMP_W + wave] + rank;   // every lane reads before the leader's update below
            kout[pos] = (uint16_t)k;
            vout[pos] = v;
        }
        if (valid && rank == 0) cnt[d * SMP_W + wave] += (unsigned)__popcll(m);
    }
    __syncthreads();
}

// The same pass for vocabularies of up to 32 x 1024 entries, entirely on chip: the wave's segment sits in registers as
// key << 16 | index (kv), the output is scattered into a 128 KB LDS buffer (a single CU issues one scattered global
// line per cycle -- 58 us per pass for 2 x 32000 stores -- while LDS takes the same scatter in ~2 us).
// DSH / DBITS: position and width of the digit inside kv. TRANSPOSE: final pass, element `pos` goes to buf[(pos % E) * 1024 + pos / E]
// so that thread t of the scan phase finds its E consecutive entries at stride 1024 (conflict-free reads).
template <int DSH, int DBITS, bool TRANSPOSE>
__device__ void radix_pass_lds(const unsigned (&kv)[SMP_E], int n, int S, int E, unsigned* buf, volatile unsigned* cnt,
                               unsigned* wtot) {
    constexpr unsigned DMASK = (1u << DBITS) - 1u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg0 = wave * S;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i = tid; i < 256 * SMP_W; i += SMP_T) cnt[i] = 0;
    __syncthreads();
    // walk 1: rank every key among the equal digits of its wave segment (ballot matching, 64 keys at a time) and count;
    // the rank (< 2048) is kept, two per register, so that the scattering walk needs no second matching
    unsigned loc[SMP_E / 2];
#pragma unroll
    for (int j = 0; j < SMP_E; j++) {
        unsigned l = 0;
        if (j * 64 < S) {
            const bool valid = seg0 + j * 64 + lane < n;
            const unsigned d = DMASK - ((kv[j] >> DSH) & DMASK);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < DBITS; b++) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const unsigned rank = (unsigned)__popcll(m & lt);
            if (valid) {
                const unsigned before = cnt[d * SMP_W + wave];   // every lane reads before the leader's update below
                l = before + rank;
                if (rank == 0) cnt[d * SMP_W + wave] = before + (unsigned)__popcll(m);
            }
        }
        if (j & 1) loc[j >> 1] |= l << 16; else loc[j >> 1] = l;
    }
    __syncthreads();
    unsigned a[4], tot = 0;                                    // exclusive scan, thread t owns counters 4t .. 4t+3
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = cnt[tid * 4 + e]; tot += a[e]; }
    unsigned incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned run = incl - tot;
    for (int w = 0; w < wave; w++) run += wtot[w];
#pragma unroll
    for (int e = 0; e < 4; e++) { cnt[tid * 4 + e] = run; run += a[e]; }
    __syncthreads();
    const bool pow2 = (E & (E - 1)) == 0;
    const int esh = 31 - __clz(E);
#pragma unroll
    for (int j = 0; j < SMP_E; j++)
        if (j * 64 < S && seg0 + j * 64 + lane < n) {
            const unsigned d = DMASK - ((kv[j] >> DSH) & DMASK);
            unsigned pos = cnt[d * SMP_W + wave] + ((loc[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
            if (TRANSPOSE) {
                const unsigned t = pow2 ? pos >> esh : pos / (unsigned)E;
                pos = (pos - t * (unsigned)E) * SMP_T + t;
            }
            buf[pos] = kv[j];
        }
    __syncthreads();
}

// fp16 inclusive prefix sum in the fixed order of the header + first index with prefix >= threshold (returned to every thread; INT_MAX: none).
// key(i): fp16 bits of the i-th probability.
// limit < n (candidate subset, see the kernel): only the first `limit` entries of the sorted order are known. A prefix value
// depends on the entries in front of it alone -- thread totals and wave totals of LATER threads never enter an earlier prefix --,
// so with the partition of the full vocabulary (E from n) every prefix below `limit` has the bits of the full scan; entries from
// `limit` on are not searched, and no hit means "not among the candidates".
// INREG: the thread's E <= 32 entries arrive in registers (pk; the caller fetched them with all loads in flight at once), key is not called
// (the two walks are chains of dependent fp16 additions: with a load in every link thread 0 alone spent 2.7 us of LDS latency on its 2 x 32 entries).
// Returns the first index whose prefix reaches the threshold (the minimum over the block: a wave minimum, then sixteen words in LDS -- `wmin` --; an LDS
// atomicMin on a generic pointer inside this shape sent hipcc / ROCm 7.2 into "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base"), or INT_MAX.
template <bool INREG, typename KeyAt>
__device__ __forceinline__ int scan_search_phase(KeyAt key, const float (&pk)[SMP_E], int n, float threshold, float* wtot, int* wmin, int limit = 0x7fffffff) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = (n + SMP_T - 1) / SMP_T;
    const int lo = tid * E, hi = min(min(n, limit), lo + E);
    float total = 0.f;
    if (INREG) {
#pragma unroll
        for (int j = 0; j < SMP_E; j++) if (lo + j < hi) total = round_h(total + pk[j]);
    } else {
        for (int i = lo; i < hi; i++) total = round_h(total + h2f(key(i, i - lo)));
    }
    float v = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(v, off);
        if (lane >= off) v = round_h(v + o);
    }
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    float base = 0.f;
    for (int w = 0; w < wave; w++) base = round_h(base + wtot[w]);
    const float prev = __shfl_up(v, 1);
    const float excl = lane > 0 ? round_h(base + prev) : base;
    float r = 0.f;
    int found = 0x7fffffff;
    if (INREG) {
#pragma unroll
        for (int j = 0; j < SMP_E; j++)
            if (lo + j < hi) {
                r = round_h(r + pk[j]);
                if (found == 0x7fffffff && round_h(excl + r) >= threshold) found = lo + j;
            }
    } else {
        for (int i = lo; i < hi; i++) {
            r = round_h(r + h2f(key(i, i - lo)));
            if (round_h(excl + r) >= threshold) { found = i; break; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(found, off); found = o < found ? o : found; }
    if (lane == 0) wmin[wave] = found;
    __syncthreads();
    int h = wmin[0];
#pragma unroll
    for (int w = 1; w < SMP_W; w++) h = wmin[w] < h ? wmin[w] : h;
    __syncthreads();                                          // (wtot / wmin may be written again by the caller's next phase)
    return h;
}

// Vocabularies of up to 32 x 1024 entries, a multiple of 8: the softmax's elementwise work (a division, an exponential) and its
// sum spread over SMX_B = 16 CUs -- on one CU it is 17 us of the sampling launch (tools/lab/sampler_kernel_time.py stamps) -- in the
// SAME arithmetic order as softmax_phase, whose canonical sum is: thread t of 1024 adds its elements t + 1024 k for ascending k, a
// wave tree over each 64 threads, a 16-lane DPP tree over the 16 wave totals. Block b of this launch is "wave b": its lanes are the
// virtual threads 64 b .. 64 b + 63. Every block finds the global maximum by itself (the whole vocabulary is 64 KB of L2 hits: no
// grid-wide exchange; max_k half(l_k / T) = half(max_k l_k / T) since the map is monotone), real thread (j, l) computes the
// elements k = 2 j, 2 j + 1 of virtual thread l, the fp32 exponentials meet in LDS, wave 0 adds each virtual thread's 32 values in
// order, and the wave tree's result goes to wave_total[b]. The fp16-rounded exponentials go to a scratch array (`logits` is still
// being read by slower blocks); topp_sample_kernel<true> normalises them with the 16-lane tree over wave_total.
constexpr int SMX_B = 16;
__global__ void __launch_bounds__(SMP_T) softmax_spread_kernel(const q4_half* __restrict__ logits, int n, float temperature,
                                                             uint16_t* __restrict__ e16, float* __restrict__ wave_total) {
    __shared__ float red[16];
    __shared__ float es[SMP_E][64];
    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, b = blockIdx.x;
    // global maximum of the raw logits, 16-byte loads (n % 8 == 0)
    float m = -3.402823466e+38f;
    for (int u = tid; u < (n >> 3); u += SMP_T) {
        const u32x4 q = reinterpret_cast<const u32x4*>(logits)[u];
#pragma unroll
        for (int d = 0; d < 4; d++) { const h2 p = as_h2(q[d]); m = fmaxf(m, fmaxf((float)p.x, (float)p.y)); }
    }
    m = block_tree_max(m, red);
    const float max_val = round_h(m / temperature);                // = max over t of half(logit_t / T), gpu_kernels.h:511,522-531
    const int E = (n + SMP_T - 1) / SMP_T;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        const int k = 2 * j + kk, t = b * 64 + lane + k * SMP_T;
        float e = 0.f;
        if (k < E && t < n) {
            float val = h2f(logits[t]);
            val /= temperature;                                    // :511
            e = expf(round_h(val) - max_val);                      // :536
            e16[t] = f2h(e);
        }
        es[k][lane] = e;                                           // (+0 for the entries past the vocabulary: x + 0 = x)
    }
    __syncthreads();
    if (j == 0) {
        float sum = 0.0f;
        for (int k = 0; k < E; k++) sum += es[k][lane];            // the virtual thread's own order
        sum = wave_sum(sum);
        if (lane == 0) wave_total[b] = sum;
    }
}

// PRE: the exponentials and the wave totals come from softmax_spread_kernel (e16 = k0, wave_total); this launch normalises and searches.
// coins != nullptr (the step is part of a captured graph): the coin of this step is coins[position] -- the host draws the
// xorshift stream (sampler.h:31-40) ahead of the replay and leaves the values in a pinned ring, so the launch carries no
// per-step argument -- else `coin`. x_next != nullptr (several steps per replay): the sampled token's embedding row becomes the
// next step's residual stream right here, like argmax_kernel does for greedy steps.
template <bool PRE>
__global__ void __launch_bounds__(SMP_T) topp_sample_kernel(q4_half* logits, int n, float temperature, int do_sort,
                                                          float coin, const float* coins, float topp, int* indices, uint16_t* k0, int* v0,
                                                          uint16_t* k1, int* v1, int* result, volatile int* pPos,
                                                          int* pPosGpu, q4_half* x_next, const q4_half* table, int dim, const float* wave_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned dyn[];   // [32768] sort buffer (on-chip path) + [4096] counters
    __shared__ float red[16];
    __shared__ unsigned wtot[16];
    __shared__ int wmin[16];
    __shared__ int s_token;
    __shared__ float red6[96];
    if (coins != nullptr) coin = coins[*pPosGpu];            // (requested first: a PCIe read that returns under the softmax)
    const float threshold = do_sort ? coin * topp : coin;    // sampler.h:57-59,69
    const bool onchip = n <= SMP_T * SMP_E;
    unsigned* buf = dyn;
    unsigned* cnt = onchip ? dyn + SMP_T * SMP_E : dyn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int hit = 0x7fffffff;                                     // (block-uniform: the value scan_search_phase returns)
    SMP_STAMP(0);
    unsigned top = 0;
    u32x4 pq[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};   // PRE: this thread's probabilities (entries 8 u .. 8 u + 7, u = tid + 1024 k), kept for the candidate path
    if (PRE) {
        // normalise (gpu_kernels.h:549): p = half(half(e) / sum), sum = the 16-lane tree over the wave totals. Elementwise and
        // order-free, so 8 consecutive entries per thread and 16-byte loads / stores
        const float sum = row16_sum(wave_total[lane & 15]);
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
            const int u = tid + kq * SMP_T;
            if (u >= (n >> 3)) break;
            const u32x4 q = reinterpret_cast<const u32x4*>(k0)[u];
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint16_t p0 = f2h(h2f((uint16_t)(q[d] & 0xffffu)) / sum), p1 = f2h(h2f((uint16_t)(q[d] >> 16)) / sum);
                o[d] = (unsigned)p0 | ((unsigned)p1 << 16);
                const unsigned t0 = (unsigned)u * 8u + 2u * d;
                const unsigned key0 = ((unsigned)p0 << 16) | (0xFFFFu - t0), key1 = ((unsigned)p1 << 16) | (0xFFFFu - (t0 + 1u));
                top = key0 > top ? key0 : top;
                top = key1 > top ? key1 : top;
            }
            reinterpret_cast<u32x4*>(logits)[u] = o;
            pq[kq] = o;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const unsigned o = __shfl_xor(top, off); top = o > top ? o : top; }
        unsigned* redu = reinterpret_cast<unsigned*>(red);
        __syncthreads();
        if (lane == 0) redu[wave] = top;
        __syncthreads();
        top = redu[0];
#pragma unroll
        for (int w = 1; w < SMP_W; w++) top = redu[w] > top ? redu[w] : top;
        __syncthreads();
    } else {
        top = softmax_phase(logits, n, temperature, indices, red);    // sampler.h:53
    }
    SMP_STAMP(5);
    const int E = (n + SMP_T - 1) / SMP_T;
    int token = 0;
    if (do_sort && top != 0 && h2f((uint16_t)(top >> 16)) >= threshold) {
        // the first sorted entry (largest probability, lowest index among equals) already reaches the threshold: its
        // prefix IS that probability, so it is the sample -- no sort, no scan (the usual case at low temperature)
        token = (int)(0xFFFFu - (top & 0xFFFFu));
    } else if (!do_sort) {                                                              // sampler.h:57-59
        const uint16_t* keys = logits;
        if (E == SMP_E && (n & 7) == 0) {
            // a thread's E = 32 consecutive keys as four 16-byte loads in flight at once (walking them with 2-byte loads is
            // 2 x 32 dependent round trips to memory: 20 us of a 40 us kernel)
            u32x4 kq[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int i0 = tid * SMP_E + c * 8;
                kq[c] = i0 < n ? *reinterpret_cast<const u32x4*>(keys + i0) : (u32x4){0u, 0u, 0u, 0u};
            }
            float pk[SMP_E];
#pragma unroll
            for (int j = 0; j < SMP_E; j++) pk[j] = h2f((uint16_t)(kq[j >> 3][(j >> 1) & 3] >> ((j & 1) * 16)));
            hit = scan_search_phase<true>([](int, int) { return (uint16_t)0; }, pk, n, threshold, red, wmin);
        } else {
            { float pk[SMP_E] = {}; hit = scan_search_phase<false>([&](int i, int) { return keys[i]; }, pk, n, threshold, red, wmin); }
        }
        if (tid == 0) token = hit == 0x7fffffff ? n - 1 : hit;                   // indices[t] == t
    } else if (onchip) {                                                         // sampler.h:60-69
        bool done = false;
        // ---- the nucleus first. The search stops at the first sorted entry whose prefix reaches the threshold, and a prefix is a function of the entries
        // in front of it: only the LARGEST probabilities matter. The entries >= 2^-12 (at most 4096 of them: they sum to <= 1; 260-739 at -t 0.5 on the
        // synthetic 7B model's logits, tools/lab/logit_stats.py) are appended, straight from the registers the normalising pass left them in and in ANY order,
        // to four lists by binary exponent -- [2^-9, 1), [2^-10, 2^-9), [2^-11, 2^-10), [2^-12, 2^-11) -- through four LDS counters. The smallest prefix of
        // lists whose mass (fp32, for choosing only) covers the threshold with a margin is the candidate set; a candidate's place in the descending stable
        // order is the size of the lists in front of its own plus the number of members of its OWN list that precede it (key, then the lower index: the
        // packed word key << 16 | 0xFFFF - index), counted in one loop over that list. Then the fp16 scan with the partition of the full vocabulary.
        // No radix pass, no ordered compaction, no second read of the probabilities; a hit among the candidates is the full sort's hit, bit for bit (the
        // cut is a key value: equal keys are complete classes). No hit, a list above 1024 entries or a flat distribution fall through to the paths below.
        if (PRE) {
            unsigned* lists = const_cast<unsigned*>(cnt);                        // [4][1024] candidates by exponent band: the radix passes' counter area, which this path
                                                                                 // does not use (the scatter below may touch every word of `buf`)
            if (tid < 4) wtot[tid] = 0u;
            __syncthreads();
            // (single appends: a wave scan of per-list counts with one atomic per wave and list was measured SLOWER, 10.5 -> 17 us for the phase -- one
            // instruction per thread of this 1024-thread block is 7 ns of the CU's issue time, and the scan's two passes over the 32 entries cost more of them
            // than the few hundred LDS atomics they replace)
#pragma unroll
            for (int kq = 0; kq < 4; kq++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned w2 = pq[kq][d];
                    if (((w2 & 0xFFFFu) >= (3u << 10)) | ((w2 >> 16) >= (3u << 10))) {       // (rare: one dword in thirty holds a candidate)
#pragma unroll
                        for (int hh = 0; hh < 2; hh++) {
                            const unsigned pb = (w2 >> (16 * hh)) & 0xffffu;
                            if (pb >= (3u << 10)) {                                      // fp16 bits of 2^-12: exponent field 3
                                const unsigned ex = pb >> 10;                            // 3 .. 15
                                const unsigned band = ex >= 6u ? 0u : 6u - ex;           // 2^-9 and above: 0; 2^-10: 1; 2^-11: 2; 2^-12: 3
                                const unsigned slot = atomicAdd(&wtot[band], 1u);
                                if (slot < (unsigned)SMP_T) lists[band * SMP_T + slot] = (pb << 16) | (0xFFFFu - ((unsigned)(tid + kq * SMP_T) * 8u + 2u * d + hh));
                            }
                        }
                    }
                }
            __syncthreads();
            const unsigned n0 = wtot[0], n1 = wtot[1], n2 = wtot[2], n3 = wtot[3];
            if (n0 <= (unsigned)SMP_T && n1 <= (unsigned)SMP_T && n2 <= (unsigned)SMP_T && n3 <= (unsigned)SMP_T) {      // (block-uniform)
                // the lists' masses: thread t adds entry t of each
                float b0 = tid < (int)n0 ? h2f((uint16_t)(lists[tid] >> 16)) : 0.f, b1 = tid < (int)n1 ? h2f((uint16_t)(lists[SMP_T + tid] >> 16)) : 0.f;
                float b2 = tid < (int)n2 ? h2f((uint16_t)(lists[2 * SMP_T + tid] >> 16)) : 0.f, b3 = tid < (int)n3 ? h2f((uint16_t)(lists[3 * SMP_T + tid] >> 16)) : 0.f;
                b0 = wave_sum(b0); b1 = wave_sum(b1); b2 = wave_sum(b2); b3 = wave_sum(b3);
                if (lane == 0) { red6[wave] = b0; red6[16 + wave] = b1; red6[32 + wave] = b2; red6[48 + wave] = b3; }
                __syncthreads();
                b0 = row16_sum(red6[lane & 15]); b1 = row16_sum(red6[16 + (lane & 15)]); b2 = row16_sum(red6[32 + (lane & 15)]); b3 = row16_sum(red6[48 + (lane & 15)]);
                const float need = threshold * 1.01f + 0.004f;
                int nb = 0;                                                      // lists taken
                if (b0 >= need) nb = 1; else if (b0 + b1 >= need) nb = 2; else if (b0 + b1 + b2 >= need) nb = 3; else if (b0 + b1 + b2 + b3 >= need) nb = 4;
                const int s1 = (int)n0, s2 = s1 + (int)n1, s3 = s2 + (int)n2, s4 = s3 + (int)n3;
                const int m = nb == 1 ? s1 : nb == 2 ? s2 : nb == 3 ? s3 : nb == 4 ? s4 : 0;
                if (m > 0 && m <= SMP_T) {
                    if (tid < m) {
                        const int band = tid < s1 ? 0 : tid < s2 ? 1 : tid < s3 ? 2 : 3;
                        const int first = band == 0 ? 0 : band == 1 ? s1 : band == 2 ? s2 : s3, cnt_b = band == 0 ? (int)n0 : band == 1 ? (int)n1 : band == 2 ? (int)n2 : (int)n3;
                        const unsigned* lb = lists + band * SMP_T;
                        const unsigned mine = lb[tid - first];
                        unsigned before = 0;
                        for (int i = 0; i < cnt_b; i++) before += lb[i] > mine ? 1u : 0u;      // (neighbouring lanes read one word: mostly an LDS broadcast)
                        const unsigned rank = (unsigned)first + before, t = rank / (unsigned)E;
                        buf[(rank - t * (unsigned)E) * SMP_T + t] = (mine & 0xFFFF0000u) | (0xFFFFu - (mine & 0xFFFFu));     // transposed for the FULL partition, as radix_pass_lds<.., true>
                    }
                    __syncthreads();
                    // the scan: m <= 1024 = the entries of threads 0 .. 31 when E = 32 (in general of the first ceil(m / E) threads) -- if those sit in
                    // wave 0, that wave alone runs scan_search_phase's arithmetic (its base is 0, no other wave's total enters): no block barrier
                    if ((m + E - 1) / E <= 64) {
                        if (wave == 0) {
                            const int lo = tid * E, hi = min(m, lo + E);
                            float pk[SMP_E];
#pragma unroll
                            for (int j = 0; j < SMP_E; j++) pk[j] = h2f((uint16_t)(buf[j * SMP_T + tid] >> 16));     // (stride 1024: conflict-free)
                            float total = 0.f;
#pragma unroll
                            for (int j = 0; j < SMP_E; j++) if (lo + j < hi) total = round_h(total + pk[j]);
                            float v = total;
#pragma unroll
                            for (int off = 1; off < 64; off <<= 1) {
                                const float o = __shfl_up(v, off);
                                if (lane >= off) v = round_h(v + o);
                            }
                            const float prev = __shfl_up(v, 1);
                            const float excl = lane > 0 ? round_h(0.f + prev) : 0.f;
                            float r = 0.f;
                            int found = 0x7fffffff;
#pragma unroll
                            for (int j = 0; j < SMP_E; j++)
                                if (lo + j < hi) {
                                    r = round_h(r + pk[j]);
                                    if (found == 0x7fffffff && round_h(excl + r) >= threshold) found = lo + j;
                                }
#pragma unroll
                            for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(found, off); found = o < found ? o : found; }
                            if (lane == 0) wmin[0] = found;
                        }
                        __syncthreads();
                        hit = wmin[0];
                    } else {
                        float pk[SMP_E];
#pragma unroll
                        for (int j = 0; j < SMP_E; j++) pk[j] = h2f((uint16_t)(buf[j * SMP_T + tid] >> 16));
                        hit = scan_search_phase<true>([](int, int) { return (uint16_t)0; }, pk, n, threshold, red, wmin, m);
                    }
                    done = hit != 0x7fffffff;
                    if (done && tid == 0) {
                        const int mi = hit, t = mi / E;
                        token = (int)(buf[(mi - t * E) * SMP_T + t] & 0xffffu);
                    }
                }
            }
            __syncthreads();                                                     // `buf`, `wtot`, `red6`, `wmin` are written again below
        }
        if (!done) {
        const int S = E * 64, seg0 = wave * S;
        unsigned kv[SMP_E];
#pragma unroll
        for (int j = 0; j < SMP_E; j++) {
            const int i = seg0 + j * 64 + lane;
            const bool valid = j * 64 < S && i < n;
            kv[j] = valid ? ((unsigned)logits[i] << 16) | (unsigned)i : 0u;
        }
        // ---- candidates first. The search stops at the first sorted entry whose prefix reaches the threshold, and a prefix is a
        // function of the entries in front of it: only the LARGEST probabilities matter. Entries >= 2^-e number at most 2^e (they sum
        // to <= 1), so the sets {p >= 2^-9}, {>= 2^-11}, {>= 2^-12} hold <= 512 / 2048 / 4096 entries: take the smallest of them whose
        // mass (fp32, for choosing only) covers the threshold with a margin, sort THOSE (two passes over <= 4 keys per thread instead
        // of 32) and scan them with the partition of the full vocabulary. A hit among them is the full sort's hit, bit for bit (equal
        // keys are complete classes: the cut is a key value); no hit -- a flat distribution, the margin too thin -- falls through to
        // the full sort below. Trained models at the CLI's temperatures end here; 92 -> ~30 us per token at vocabulary 32000.
        {
            constexpr unsigned C0 = (15u - 9u) << 26, C1 = (15u - 11u) << 26, C2 = (15u - 12u) << 26;   // fp16 2^-9, 2^-11, 2^-12 << 16
            float m0 = 0.f, m1 = 0.f, m2 = 0.f;
            unsigned c012 = 0;                                                   // three 10-bit counts (<= 32 each per thread)
#pragma unroll
            for (int j = 0; j < SMP_E; j++) {
                const float pj = h2f((uint16_t)(kv[j] >> 16));
                const bool a = kv[j] >= C0, b = kv[j] >= C1, c = kv[j] >= C2;
                m0 += a ? pj : 0.f; m1 += b ? pj : 0.f; m2 += c ? pj : 0.f;
                c012 += (a ? 1u : 0u) + (b ? 1u << 10 : 0u) + (c ? 1u << 20 : 0u);
            }
            // one barrier for the six block totals (the masses only choose the cut: their summation order is free; the counts
            // are integers below 2^24, exact in fp32)
            float f0 = (float)(c012 & 1023u), f1 = (float)((c012 >> 10) & 1023u), f2 = (float)(c012 >> 20);
            m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2); f0 = wave_sum(f0); f1 = wave_sum(f1); f2 = wave_sum(f2);
            if (lane == 0) { red6[wave] = m0; red6[16 + wave] = m1; red6[32 + wave] = m2; red6[48 + wave] = f0; red6[64 + wave] = f1; red6[80 + wave] = f2; }
            __syncthreads();
            m0 = row16_sum(red6[lane & 15]); m1 = row16_sum(red6[16 + (lane & 15)]); m2 = row16_sum(red6[32 + (lane & 15)]);
            f0 = row16_sum(red6[48 + (lane & 15)]); f1 = row16_sum(red6[64 + (lane & 15)]); f2 = row16_sum(red6[80 + (lane & 15)]);
            const float need = threshold * 1.01f + 0.004f;
            unsigned cut = 0; int m = 0;
            if (m0 >= need) { cut = C0; m = (int)f0; } else if (m1 >= need) { cut = C1; m = (int)f1; } else if (m2 >= need) { cut = C2; m = (int)f2; }
            if (m > 0 && m <= 4096) {
                // ordered compaction (index order inside and across the wave segments: the sort passes are stable, equal keys keep
                // ascending index) into the upper half of the sort buffer
                unsigned* cand = buf + SMP_T * SMP_E / 2;
                const unsigned long long lt = (1ull << lane) - 1ull;
                float mine = 0.f;                                                // this wave's candidates (counted per lane, then summed:
#pragma unroll                                                                   //  32 ballot masks kept alive for the walk below would spill)
                for (int j = 0; j < SMP_E; j++) mine += (kv[j] >= cut && j * 64 < S && seg0 + j * 64 + lane < n) ? 1.f : 0.f;
                mine = wave_sum(mine);
                if (lane == 0) wtot[wave] = (unsigned)mine;
                __syncthreads();
                unsigned off = 0;
                for (int w = 0; w < wave; w++) off += wtot[w];
#pragma unroll
                for (int j = 0; j < SMP_E; j++) {
                    const bool take = kv[j] >= cut && j * 64 < S && seg0 + j * 64 + lane < n;
                    const unsigned long long bm = __ballot(take);
                    if (take) cand[off + (unsigned)__popcll(bm & lt)] = kv[j];
                    off += (unsigned)__popcll(bm);
                }
                __syncthreads();
                const int E2 = (m + SMP_T - 1) / SMP_T, S2 = E2 * 64, sg = wave * S2;   // <= 4 keys per thread
                unsigned kc[SMP_E];
#pragma unroll
                for (int j = 0; j < SMP_E; j++) kc[j] = (j * 64 < S2 && sg + j * 64 + lane < m) ? cand[sg + j * 64 + lane] : 0u;
                __syncthreads();                                                 // the candidates are in registers: `buf` is free
                radix_pass_lds<16, 7, false>(kc, m, S2, E2, buf, cnt, wtot);
#pragma unroll
                for (int j = 0; j < SMP_E; j++) kc[j] = (j * 64 < S2 && sg + j * 64 + lane < m) ? buf[sg + j * 64 + lane] : 0u;
                __syncthreads();
                radix_pass_lds<23, 8, true>(kc, m, S2, E, buf, cnt, wtot);       // transposed for the FULL partition (E entries per thread)
                __syncthreads();
                { float pk[SMP_E];
#pragma unroll
                  for (int j = 0; j < SMP_E; j++) pk[j] = h2f((uint16_t)(buf[j * SMP_T + tid] >> 16));     // (a thread's E consecutive entries: stride 1024, conflict-free)
                  hit = scan_search_phase<true>([](int, int) { return (uint16_t)0; }, pk, n, threshold, red, wmin, m); }
                done = hit != 0x7fffffff;                                        // (block-uniform: read after the phase's last barrier)
                if (done && tid == 0) {
                    const int mi = hit, t = mi / E;
                    token = (int)(buf[(mi - t * E) * SMP_T + t] & 0xffffu);
                }
                __syncthreads();                                                 // `hit` and `buf` are re-used below
            }
        }
        if (!done) {
        radix_pass_lds<16, 7, false>(kv, n, S, E, buf, cnt, wtot);   // key bits 0-6
#pragma unroll
        for (int j = 0; j < SMP_E; j++) {
            const int i = seg0 + j * 64 + lane;
            kv[j] = (j * 64 < S && i < n) ? buf[i] : 0u;
        }
        __syncthreads();                                                         // all segments are in registers
        radix_pass_lds<23, 8, true>(kv, n, S, E, buf, cnt, wtot);     // key bits 7-14 (bit 15 is the sign: 0)
        { float pk[SMP_E];
#pragma unroll
                  for (int j = 0; j < SMP_E; j++) pk[j] = h2f((uint16_t)(buf[j * SMP_T + tid] >> 16));     // (a thread's E consecutive entries: stride 1024, conflict-free)
                  hit = scan_search_phase<true>([](int, int) { return (uint16_t)0; }, pk, n, threshold, red, wmin); }
        if (tid == 0) {
            const int mi = hit == 0x7fffffff ? n - 1 : hit;                      // gpu_kernels.h:560,574
            const int t = mi / E;
            token = (int)(buf[(mi - t * E) * SMP_T + t] & 0xffffu);
        }
        }
        }
    } else {
        radix_pass(logits, nullptr, k0, v0, n, 0, cnt, wtot);
        radix_pass(k0, v0, k1, v1, n, 8, cnt, wtot);
        const uint16_t* keys = k1;
        { float pk[SMP_E] = {}; hit = scan_search_phase<false>([&](int i, int) { return keys[i]; }, pk, n, threshold, red, wmin); }
        if (tid == 0) token = v1[hit == 0x7fffffff ? n - 1 : hit];
    }
    SMP_STAMP(6);
    if (x_next != nullptr) {                                 // (uniform: a kernel argument)
        if (tid == 0) s_token = token;
        __syncthreads();
        const int tk = s_token;
        for (int u = tid; u < (dim >> 3); u += SMP_T)
            reinterpret_cast<u32x4*>(x_next)[u] = reinterpret_cast<const u32x4*>(table + (size_t)tk * dim)[u];
    }
    if (tid == 0) {
        int token_pos = *pPosGpu;                            // == *pPos (:579-580) without the PCIe read
        token_pos++;
        result[token_pos] = token;                           // :578
        __threadfence_system();                              // the host may be spinning on *pPos (q4_wait_pos)
        *pPos = token_pos;
        *pPosGpu = token_pos;
    }
}

}  // namespace

// scratch and LDS opt-in of the sampling launch: outside any stream capture (hipMalloc / hipFuncSetAttribute are not capturable)
extern "C" __attribute__((visibility("hidden"))) int q4_sample_topp_prepare(Sampler* sampler) {
    const int n = sampler->vocab_size;
    if (sampler->temp_storage_bytes_scan == 0) {                                    // (the scan scratch of sampler.h:72-76: 16 wave totals here)
        Q4_HIP(hipMalloc(&sampler->tempStorage_scan, 256));
        sampler->temp_storage_bytes_scan = 256;
    }
    if (sampler->temp_storage_bytes_sort == 0) {                                    // sampler.h:62-66 (lazy scratch)
        const size_t bytes = 2 * ((size_t)n * sizeof(uint16_t) + 256) + 2 * ((size_t)n * sizeof(int) + 256);
        Q4_HIP(hipMalloc(&sampler->tempStorage_sort, bytes));
        sampler->temp_storage_bytes_sort = bytes;
    }
    const int rc = q4::lds_opt_in((const void*)topp_sample_kernel<false>, SMP_T * SMP_E * 4 + 256 * SMP_W * 4);      // (once per device)
    return rc ? rc : q4::lds_opt_in((const void*)topp_sample_kernel<true>, SMP_T * SMP_E * 4 + 256 * SMP_W * 4);
}

extern "C" __attribute__((visibility("hidden"))) int q4_sample_topp_device(Sampler* sampler, RunState* s, float coin, const float* coins,
                                                                           q4_half* x_next, const q4_half* table, int dim) {
    const int n = sampler->vocab_size;
    const int do_sort = !(sampler->topp <= 0 || sampler->topp >= 1);
    if (x_next != nullptr && (dim & 7)) return Q4_ERR_UNSUPPORTED_SIZE;
    if (coins == nullptr) { int rc = q4_sample_topp_prepare(sampler); if (rc) return rc; }   // (graph path: prepared before the capture)
    char* base = (char*)sampler->tempStorage_sort;
    const size_t kb = ((size_t)n * sizeof(uint16_t) + 255) / 256 * 256, vb = ((size_t)n * sizeof(int) + 255) / 256 * 256;
    uint16_t* k0 = (uint16_t*)base; uint16_t* k1 = (uint16_t*)(base + kb);
    int* v0 = (int*)(base + 2 * kb); int* v1 = (int*)(base + 2 * kb + vb);
    const size_t smem = (n <= SMP_T * SMP_E ? (size_t)SMP_T * SMP_E * 4 : 0) + 256 * SMP_W * 4;
    float* wave_total = (float*)sampler->tempStorage_scan;
    // (the spread softmax takes max(logit) / T for the maximum of the scaled logits: true for T > 0 only. A negative temperature -- the CLI clamps it,
    // the C / Python API does not -- goes the one-block way, which like the reference takes the maximum of the scaled values and stays finite)
    if (n <= SMP_T * SMP_E && (n & 7) == 0 && sampler->temperature > 0.0f) {      // the softmax over 16 CUs, then normalise + search on one
        Q4_LAUNCH(softmax_spread_kernel, dim3(SMX_B), dim3(SMP_T), 0, s->logits, n, sampler->temperature, k0, wave_total);
        Q4_LAUNCH(topp_sample_kernel<true>, dim3(1), dim3(SMP_T), smem, s->logits, n, sampler->temperature, do_sort, coin, coins, sampler->topp,
                  sampler->indices, k0, v0, k1, v1, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos, x_next, table, dim, wave_total);
    } else {
        Q4_LAUNCH(topp_sample_kernel<false>, dim3(1), dim3(SMP_T), smem, s->logits, n, sampler->temperature, do_sort, coin, coins, sampler->topp,
                  sampler->indices, k0, v0, k1, v1, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos, x_next, table, dim, wave_total);   // :53-80
    }
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
