// gemv_strip.h -- "strips": the fused gate/up GEMV (rmsnorm_kernel + ffn_matvec_silu_kernel, gpu_kernels.h:72-105, 256-275) on LDS-DMA rings. Ships
// for wide matrices (ffn_strip_covers below); the profiling build runs every variant anywhere the shape is covered
// (q4_set_gemv_early(11, 8..14); -1 = the wave-owned kernel always). NO loader wave: one 16-wave block per CU owns a contiguous range of columns;
// every wave streams its OWN units -- unit u = wv + 16 i of the block's (column, matrix) pairs, 2 KiB each at K = 4096, 2.5 KiB at K = 5120 -- with
// `buffer_load_dwordx4 ... nt lds` into a private ring of D 1 KiB pieces, waits for its oldest piece with vmcnt (a wave's loads return in order),
// reads it back with ds_read_b128, re-issues and multiplies with the denormal-nibble v_dot2c body of gemv_q4.h. Four waves per SIMD (the loader /
// consumer engine of gemv_engine.hip had two), x staged once per CU by waves 0..7 (0..9) and held in 32 (48) registers by every wave. Same arithmetic
// in the same order as gemv_q4_kernel<MODE_FFN>, bit for bit (tests/prof_cases.py).
//   MODE 0 ("plain"): a wave sends its whole ring at entry and re-issues an entry when it has read it: D pieces in flight per wave.
//   MODE 1 ("paced"): at most TWO pieces of a wave in flight whatever the depth of its ring (32 KiB per CU is what the CU's memory
//          pipe takes without stalling the issue; more in flight measured slower): an issue is
//          preceded by vmcnt(1). The waves that do not stage x fill their rings during the x chain, one piece per piece landed;
//          the x chain synchronises through LDS counters (a wave stalled in vmcnt must not hold a hardware barrier up).
//   MODE 2: MODE 1 with the dealing order rotated by eight waves: the waves that do not stage x take the longer share.
#pragma once
#include "gemv_q4.h"
#include "lds_dma.h"

namespace q4 {

// Flag words in LDS. A wave's LDS operations execute in program order, so "write data, then bump the flag" and "see the flag, then
// read data" need no hardware fence inside a workgroup; the relaxed forms + compiler barriers keep hipcc from re-ordering them AND
// from attaching its own waits: for an acquire / release at workgroup scope it emits s_waitcnt vmcnt(0) here (it cannot see the
// asm LDS-DMA pieces, but it counts the x loads at the kernel's entry), which would drain the loader's stream at every flag.
__device__ __forceinline__ unsigned lds_peek(unsigned* p) {
    const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_post(unsigned* p, unsigned v, unsigned lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_bump(unsigned* p, unsigned lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// Every wait is on a wave of the SAME block (all resident by construction: no scheduling order can wedge it), and still bounded
// (~10 s): a wait that runs out -- a logic error, not a race -- raises the block's fail word, and the block then stores NaN, so the
// failure is loud in every consumer of the result instead of a hung GPU or plausible garbage.
constexpr unsigned ENG_SPIN_LIMIT = 1u << 27;
__device__ __forceinline__ unsigned lds_wait_ge(unsigned* p, unsigned target, unsigned* fail) {
    unsigned v = lds_peek(p);
    for (unsigned n = 0; v < target; n++) {
        if (n >= ENG_SPIN_LIMIT || ((n & 1023u) == 1023u && lds_peek(fail) != 0u)) { lds_post(fail, 1u, 0u); break; }
        __builtin_amdgcn_s_sleep(1);
        v = lds_peek(p);
    }
    return v;
}

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
    static constexpr unsigned STAMP = TOT + 512u;                            // [64] wall-clock stamps (STAMPS builds)
    static constexpr unsigned FLAGS = STAMP + 512u;                          // paced builds: sum-of-squares arrivals, staged arrivals, fail
    static constexpr unsigned BYTES = FLAGS + 64u;
};
enum { SF_SS = 0, SF_STAGED = 1, SF_FAIL = 2 };

// stamps (tools/timeline_strip.py; kept in LDS and written out at the end: a global store would count in vmcnt): [0] wave 0 entry,
// [1] its x landed, [2] sum of squares exchanged, [3] x staged, [4 + i] its unit i multiplied, [12] its totals written, [13] outputs
// stored; [16 + w] wave w entry, [32 + w] wave w's first piece read, [48 + w] wave w's last unit multiplied
#define SSTAMP(k) do { if (STAMPS && lane == 0) st[(k)] = wall_clock64(); } while (0)
template <bool NORM, int D, int MODE, bool STAMPS, int TS = 2>
__global__ void __launch_bounds__(STRIP_WAVES * 64) ffn_strip_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w0, const void* arg_w1, const unsigned wbytes,
                                                                     const unsigned cbase, const unsigned crem, const GemvArgs a) {
    // the scalars the entry needs come first: built with -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave instead of
    // through a scalar load from the kernel-argument segment (tools/timeline_strip.py: "x landed")
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
        if (p < (int)L::NS_S) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].s, 0, a.N * a.sh * 2, 0x00020000);
            dma_piece_default(L::SIDE_S + m * L::SIDE_S_BYTES + p * 1024u, voff, rs, c0 * (G * 2u) + p * 1024u);
        } else {
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].z, 0, a.N * a.pzh * 4, 0x00020000);
            dma_piece_default(L::SIDE_Z + m * L::SIDE_Z_BYTES + (p - (int)L::NS_S) * 1024u, voff, rz, c0 * (ZW * 4u) + (p - (int)L::NS_S) * 1024u);
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
        if (NORM) v = rms_apply8(v, wraw, ss);
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
        const float total = reduce4_rows(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row r: unit g4 * 4 + r
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
        if (p < (int)L::NS_S) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].s, 0, a.N * a.sh * 2, 0x00020000);
            dma_piece_default(L::SIDE_S + m * L::SIDE_S_BYTES + p * 1024u, voff, rs, c0 * (G * 2u) + p * 1024u);
        } else {
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)a.m[m].z, 0, a.N * a.pzh * 4, 0x00020000);
            dma_piece_default(L::SIDE_Z + m * L::SIDE_Z_BYTES + (p - (int)L::NS_S) * 1024u, voff, rz, c0 * (ZW * 4u) + (p - (int)L::NS_S) * 1024u);
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
        if (NORM) v = rms_apply8(v, wraw, ss);
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
                    asm volatile("" : "+v"(cA), "+v"(cB));                  // the piece's arithmetic stays in front of the next piece's wait (section 3.1b)
                }
                cs[2 * r] = cA;
                cs[2 * r + 1] = cB;
            }
        }
        const float total = reduce4_rows(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;   // row: pair (2 g2 + row / 2), column row % 2
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
// each other: on a CU-masked stream the form would be correct, only slow). g_engine: 0 = the product's choice -- strips from
// STRIP_MIN_COLS columns per CU on, where the wave-owned kernel's grid needs a seventh row of blocks per CU and strips measure
// faster (tools/sweep_strips.py, per launch in a graph: 12800 columns 11.83 -> 10.56 us, 13312 11.89 -> 10.96, the Mistral / Llama-3
// hidden size 14336 12.4 -> 11.45, 885 -> 902 tokens/s on the Mistral-7B geometry; up to 48 columns per CU -- Llama-2-7B's 11008 is 43 --
// the two were level at first: the threshold was 49 until the strips' kernel arguments were preloaded (csrc/Makefile GEMVFLAGS) and their last
// piece but one stopped draining. With both, per launch in a graph, wave-owned | strips: 16 columns per CU 4.92 | 5.39 us, 20: 5.88 | 5.92, 24: 6.25 |
// 6.46, 28: 7.12 | 6.98, 32: 7.78 | 7.75, 36: 8.57 | 8.23, 40: 9.15 | 9.01, 43 (Llama-2-7B): 10.12 | 9.69, 46: 10.08 | 9.72; Llama-2-7B -n 256
// 982.7 -> 990.8, -n 2048 872.6 -> 879.5 tokens/s (tools/ab.py against a build with the threshold at 43)) --; the profiling build also takes
// -1 = never, 8 .. 14 = this variant wherever the shape is covered.
#ifdef Q4_STRIP_MIN_COLS
constexpr int STRIP_MIN_COLS = Q4_STRIP_MIN_COLS;   // (make exp EXPFLAGS_GEMV=-DQ4_STRIP_MIN_COLS=49: the wave-owned kernel for Llama-2-7B's gate/up, for the A/B)
#else
constexpr int STRIP_MIN_COLS = 36;
#endif
extern int g_engine;
static inline bool strip_k5120(const GemvArgs& a) { return a.K == 5120 && a.pw4 == 160 && a.sh == 40 && a.pzh == 5; }
static bool ffn_strip_shape(const GemvArgs& a) {
    const int nb = cu_count();
    // K = 4096, or K = 5120 (13B) where the wave-owned kernel would run its shared half slot, whose arithmetic the strips repeat
    const bool k = (a.K == 4096 && a.pw4 == 128 && a.sh == 32 && a.pzh == 4) || (strip_k5120(a) && a.nslots == 3 && half_tail(a));
    return k && a.N / nb >= 16 && divUp(a.N, nb) <= STRIP_NCMAX && stream_cu_count() == nb;
}
static bool ffn_strip_covers(const GemvArgs& a) {
    if (g_engine == 0 || g_engine == 15 || g_engine == 19) return a.N / cu_count() >= STRIP_MIN_COLS && g_ablate == 0 && ffn_strip_shape(a);   // (15: profiling, the product's gate/up choice without the down-projection strips)
#ifdef Q4_PROFILING
    return g_engine >= 8 && g_engine <= 14 && g_engine != 11 && g_ablate == 0 && ffn_strip_shape(a);
#else
    return false;
#endif
}
template <bool NORM, int D, int MODE, bool STAMPS, int TS = 2>
static int launch_strip(const GemvArgs& a) {
    constexpr size_t smem = StripLds<D, TS>::BYTES;
    if (smem > 64 * 1024) {    // (the product's D = 2 form needs 54 KiB: no opt-in, nothing that a graph capture could not record)
        static bool opted = false;
        if (!opted) {
            Q4_HIP(hipFuncSetAttribute((const void*)ffn_strip_kernel<NORM, D, MODE, STAMPS, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            opted = true;
        }
    }
    const unsigned nb = (unsigned)cu_count();
    Q4_LAUNCH((ffn_strip_kernel<NORM, D, MODE, STAMPS, TS>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
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
#ifdef Q4_PROFILING
template <bool NORM, bool STAMPS>
static int launch_strip_setting(const GemvArgs& a) {
    switch (g_engine) {
        case 9: return launch_strip<NORM, 4, 0, STAMPS>(a);
        case 10: return launch_strip<NORM, 8, 0, STAMPS>(a);
        case 12: return launch_strip<NORM, 4, 1, STAMPS>(a);
        case 13: return launch_strip<NORM, 8, 1, STAMPS>(a);
        case 14: return launch_strip<NORM, 8, 2, STAMPS>(a);
        default: return launch_strip<NORM, 2, 0, STAMPS>(a);   // 0 (the product's form), 8
    }
}
static int launch_ffn_strip(const GemvArgs& a) {
    const bool norm = a.rms_w != nullptr;
    if (g_engine != 19 && g_engine != 9 && strip_pairs(a)) return launch_strip_pair(a);   // (19: the column-unit form of K = 5120, for the A/B)
    if (strip_k5120(a)) {
        if (g_engine == 9) return norm ? launch_strip<true, 4, 0, false, 3>(a) : launch_strip<false, 4, 0, false, 3>(a);
        return norm ? launch_strip<true, 2, 0, false, 3>(a) : launch_strip<false, 2, 0, false, 3>(a);
    }
    if (a.dbg) return norm ? launch_strip_setting<true, true>(a) : launch_strip_setting<false, true>(a);
    return norm ? launch_strip_setting<true, false>(a) : launch_strip_setting<false, false>(a);
}
#else
static int launch_ffn_strip(const GemvArgs& a) {
    if (strip_pairs(a)) return launch_strip_pair(a);
    if (strip_k5120(a)) return a.rms_w ? launch_strip<true, 2, 0, false, 3>(a) : launch_strip<false, 2, 0, false, 3>(a);
    return a.rms_w ? launch_strip<true, 2, 0, false>(a) : launch_strip<false, 2, 0, false>(a);
}
#endif

}  // namespace q4
