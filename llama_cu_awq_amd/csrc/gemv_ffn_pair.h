// gemv_ffn_pair.h -- the FFN half of a layer as ONE launch (fusion level 4): rmsnorm + gate/up + SiLU (rmsnorm_kernel + ffn_matvec_silu_kernel,
// gpu_kernels.h:72-105, 256-275; llama2_q4.cu:326-329) and the down projection with its residual add (mat_vec_kernel_int4, gpu_kernels.h:213-240;
// llama2_q4.cu:332). One 16-wave block per CU, all blocks resident (a CU holds one: 146 KiB of LDS):
//   phase 1  the block's share of gate/up columns exactly as gemv_strip.h streams them (every wave its own (column, matrix) units on a private ring
//            of two 1 KiB LDS-DMA pieces). A block owns whole column PAIRS, so that its slice of hb leaves as 8-byte granules {two halves, tag}.
//   seam     a wave's stream does not end with its last gate/up piece: it continues with the block's share of the DOWN projection's weights -- the
//            16 output columns of this CU are one contiguous run of the tensor, 86 KiB at Llama-2-7B -- which land in LDS the gate/up phase does not
//            use and stay there: the down stream runs under the seam instead of behind a launch boundary. Wave 0 publishes the block's slice of hb
//            (one sc1 store per lane; also the plain hb vector, RunState::hb); waves 1..15 gather the whole vector -- every thread its own one or two
//            8-half chunks, re-reading only what does not carry the launch's tag yet, one more down piece between two passes -- and stage it in the
//            K-split kernel's permuted layout over the dead gate/up rings.
//   phase 2  the down projection's dot products on weights that are already in LDS: units (column, k-part), lanes, slots and the part-0-then-part-1
//            sum of gemv_q4_kernel<MODE_PLAIN, 3, 4, false, 0, 2> (= down_strip_kernel<3, false>), residual add, one rounding.
//   phase 3  (fusion level 5, template QKV) rmsnorm + q/k/v + RoPE + KV write of the NEXT layer (gemv_q4_kernel<MODE_QKV, 2, 4, true>, gpu_kernels.h:72-105,
//            242-254, 332-355; llama2_q4.cu:300-317): the block's 48 q / k / v columns (eight RoPE pairs of each matrix) stream into the LDS the down weights
//            have left, one request in front of each of phase 2's dot-product steps; the block's 16 new values of x leave as granules, waves 0 .. 7 gather the
//            vector and run the QKV kernel's x chain (512 chunk partials, canonical reduction, norm folded into the staging) while every wave unpacks its
//            pieces; dot products on resident weights, the RoPE epilogue from the rotation table, q and the K / V rows of the position; one thread
//            advances the epoch for the attention -> o-proj launch behind it.
// Same arithmetic in the same order as the launches it replaces: bit for bit (tests/test_ffn_pair_gpu.py, tests/prof_cases.py).
// Protocol (as layer_attn.h): the tag is the model's epoch word, advanced by the PRECEDING fused QKV launch; every block is producer first and
// consumer second, producers wait for nobody, so the launch cannot wedge while all blocks are resident (grid = CUs of the device, the stream unmasked:
// ffn_pair_covers); every wait is bounded and a run-out sets the model's sticky error word (q4_handoff_status: one clean retry at fusion level 1).
#pragma once
#include <type_traits>
#include "gemv_strip.h"
#include "attention.h"

namespace q4 {

struct FfnPairArgs {
    GemvMat d;                 // the down projection's tensors
    q4_half* xio;              // residual stream: the launch's rmsnorm input, and x += down at its end (llama2_q4.cu:332, accum)
    const unsigned* sync;      // hand-off words of the model: [SYNC_ERROR], [SYNC_EPOCH] (one aligned 8-byte word)
    unsigned* error;           // = sync + SYNC_ERROR
    u32x2v* gran;              // hidden / 2 granules {hb[2 g], hb[2 g + 1], tag}
    int Kd, Nd, pw4, pzh, sh, ku;   // down projection: K = hidden, N = dim, packed geometry, uint4 units per k-part (gemv_q4.h, KS = 2)
    unsigned dbase, drem;      // its output columns per block
    unsigned pre;              // gather mode: bit 0 = the first pass with plain loads, bit 1 = ... once the wave's own down pieces have landed
    unsigned tag_add;          // 0 in the network. bench launches without a QKV launch in between (q4_bench.hip) make their tags distinct with it
    int mute;                  // profiling build: the blocks do not publish (a real time-out, tests/prof_cases.py)
    unsigned long long* dbg;   // profiling build: [block][64] wall-clock stamps
};

// phase 3 (fusion level 5): rmsnorm + q/k/v + RoPE + KV write of the NEXT layer (the launch then replaces gemv_q4_kernel<MODE_QKV, 2, 4, true> as well)
struct QkvNextArgs {
    GemvMat m[3];              // q, k, v of the next layer (K = N = dim = 4096, multi-head: kv_dim == dim)
    const q4_half* rms_w;      // its rms_att_weight
    q4_half* q;                // RunState::q
    q4_half* kc;               // key / value cache of the next layer (its layer offset applied by the host)
    q4_half* vc;
    const int* pPos;
    const float2* rope_table;  // [seq_len][head_size / 2] (cos, sin)
    u32x2v* xgran;             // dim / 2 granules {x[2 g], x[2 g + 1], tag}: the residual stream between phases 2 and 3
    unsigned* bump;            // the epoch word of the attention -> o-proj launch that follows (advanced once, at the very end, by block 0)
    int head_size, kv_dim;
    unsigned ppb;              // RoPE pairs (i, i + head_size / 2) per block and matrix: dim / 2 / blocks (8)
};

// phases A and O (fusion level 6, template ATT): MultiHeadAttention and the output projection with its residual add of THIS layer (llama2_q4.cu:320-323) in front
// of phase 1 -- the launch is then the whole layer behind its q / k / v, and with QKV the next layer's q / k / v as well
struct LayerAttArgs {
    AttArgs att;               // output = RunState::xb, q, this layer's cache rows, the bin
    GemvMat o;                 // the output projection's tensors (K = N = dim = 4096)
    u32x2v* agran;             // dim / 2 granules {xb[2 g], xb[2 g + 1], tag}: the attention output between phases A and O
    u32x2v* xogran;            // dim / 2 granules of the residual stream between phases O and 1
    unsigned nheads, natt;     // heads; blocks [0, natt) run a (head, V slice) unit of the attention (natt = 4 x heads <= blocks)
};

constexpr int FP_ROWS = 6;             // 64-unit rows of the staged hb vector: hidden <= 12288
constexpr int FP_NC2MAX = 16;          // down projection columns per block
struct FfnPairLds {
    using G = StripLds<2, 2>;
    static constexpr unsigned XS2 = 0;                              // [6][4][64] x 16 B permuted hb -- over the gate/up rings, dead behind barrier A
    static constexpr unsigned SX2 = XS2 + FP_ROWS * 4096u;          // [6][64] -(sum of the 32 inputs) * 2^-20
    static_assert(SX2 + FP_ROWS * 256u <= G::SIDE_S, "the staged hb vector fits in the gate/up rings");
    static constexpr unsigned TOT2 = G::BYTES;                      // [16][2] k-part totals
    static constexpr unsigned DSIDE_S = TOT2 + 256u;                // 3 KiB: 16 columns x <= 96 groups x 2 B
    static constexpr unsigned DSIDE_Z = DSIDE_S + 3072u;            // 1 KiB: 16 columns x <= 16 words x 4 B
    static constexpr unsigned DW = DSIDE_Z + 1024u;                 // the block's columns of the down projection, as they lie in the tensor
    static constexpr unsigned DW_BYTES = 96u * 1024u;               // (phase 3 keeps the block's 48 q / k / v columns of 2 KiB here once the down weights are unpacked)
    static constexpr unsigned QSIDE_S = DW + DW_BYTES;              // 3 KiB: 48 columns x 32 groups x 2 B
    static constexpr unsigned QSIDE_Z = QSIDE_S + 3072u;            // 1 KiB: 48 columns x 4 words x 4 B
    static constexpr unsigned STAMP2 = QSIDE_Z + 1024u;             // profiling build: [64] more stamps ([0..15] phase 2 done per wave, [16..31] phase 3 done per wave, [32..] the second seam)
    static constexpr unsigned BYTES = STAMP2 + 512u;
    // phases A / O (fusion level 6), before phase 1: the attention role's scratch lies over the x staging area; the DW region takes what is requested AHEAD of
    // phase 1 -- three gate/up pieces a wave (pieces 2 .. 4) where the down columns 0 .. 7 will land (and behind the last column where they do not fit there), and the down
    // projection's columns 8 .. 15 in their final place; [TOT2 + 128, + 32): the residual of the block's down columns
    static constexpr unsigned ATT_LDS = G::XS;                     // (32 + 16 x 128 + 256) floats <= XS .. TOT
    static constexpr int NCREDIT = 3;
    // 1 KiB entry k = 16 (piece - 2) + wave of the pieces ahead: below the down columns 8 .. 15 where it fits, behind them otherwise
    static __host__ __device__ constexpr unsigned credit_slot(unsigned k, unsigned colbytes2) {
        const unsigned lowcap = (8u * colbytes2) >> 10, tail0 = (16u * colbytes2 + 1023u) >> 10;
        return DW + (k < lowcap ? k : tail0 + (k - lowcap)) * 1024u;
    }
    static __host__ __device__ constexpr bool credit_fits(unsigned colbytes2) { return credit_slot(16u * NCREDIT - 1u, colbytes2) + 1024u <= DW + DW_BYTES; }
};
static_assert(FfnPairLds::BYTES <= 160 * 1024, "one block per CU");
static_assert((32 + 16 * 128 + 256) * 4 <= FfnPairLds::G::STAMP - FfnPairLds::G::XS, "phase A / O layout");
static_assert(FfnPairLds::QSIDE_S % 16 == 0, "alignment");
static_assert(FfnPairLds::DW % 16 == 0 && FfnPairLds::TOT2 % 16 == 0, "alignment");
constexpr unsigned FP_POLL_LIMIT = 1u << 18;    // gather passes without a down piece in between (each a memory round trip + s_sleep): ~0.3 s, then give up

#define FPSTAMP(k) do { if (STAMPS && lane == 0) st[(k)] = wall_clock64(); } while (0)
#define FPSTAMP2(k) do { if (STAMPS && lane == 0) reinterpret_cast<unsigned long long*>(smem + P::STAMP2)[(k)] = wall_clock64(); } while (0)
template <bool NORM, bool STAMPS, bool QKV = false, int ATT = 0>
__global__ void __launch_bounds__(STRIP_WAVES * 64) ffn_pair_kernel(const u32x4* __restrict__ arg_x, const u32x4* __restrict__ arg_rms, const void* arg_w0, const void* arg_w1, const unsigned wbytes,
                                                                    const unsigned pbase, const unsigned prem, const GemvArgs a, const FfnPairArgs p, const QkvNextArgs q3, const LayerAttArgs la) {
    constexpr int TS = 2, D = 2;
    constexpr unsigned CB = 2048u, G = 32u, ZW = 4u;
    constexpr int NSTAGE = 8;
    using L = StripLds<D, TS>;
    using P = FfnPairLds;
    static_assert(!ATT || NORM, "whole-layer form: the FFN half normalises");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned bp = blockIdx.x * pbase + (blockIdx.x < prem ? blockIdx.x : prem);
    const unsigned c0 = 2u * bp;                                   // first gate/up column of this block (even)
    const int nc = 2 * (int)(pbase + (blockIdx.x < prem ? 1u : 0u));
    const unsigned c0d = blockIdx.x * p.dbase + (blockIdx.x < p.drem ? blockIdx.x : p.drem);   // first down column
    const int nc2 = (int)(p.dbase + (blockIdx.x < p.drem ? 1u : 0u));
    // gate/up units: u = gw + 16 i, gw = wave as in gemv_strip.h: where the units do not divide by sixteen the OLDEST waves carry one more -- they also win the
    // SIMD's issue arbitration. Dealt the other way round (gw = 15 - wave) the block's last wave finished 1.8 us later (tools/lab/ffn_pair_check.py)
    const int gw = wave;
    const int mat = gw & 1;
    const int nu = (2 * nc - gw + 15) >> 4;
    const int npieces = TS * nu;
    const unsigned voff = lane * 16u;
    const bool stager = wave < NSTAGE;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(smem + L::STAMP);
    if (STAMPS && wave == 0) { st[lane] = 0ull; reinterpret_cast<unsigned long long*>(smem + P::STAMP2)[lane] = 0ull; FPSTAMP(0); }     // [0] entry, [1..3] x chain, [4] barrier A, [5] published, [6..7] wave 1 first pass / complete, [9] barrier B, [10] dots, [11] stored,
                                                                  // [12] passes; per wave: [16 + w] gathered, [32 + w] gate/up done, [48 + w] down pieces landed

    // ---- entry, as ffn_strip_kernel: x first, then the hand-off word, the side data of both phases, then the ring
    // (whole-layer form: x does not exist yet -- the attention role's rows first, or nothing in front of the o-proj weights)
    u32x4 xraw = {0u, 0u, 0u, 0u}, wraw = {0u, 0u, 0u, 0u};
    if (!ATT && stager) {
        const u32x4* px = arg_x + tid;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xraw) : "v"(px) : "memory");
        if (NORM) {
            const u32x4* pw = arg_rms + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wraw) : "v"(pw) : "memory");
        }
    }
    unsigned resid_raw = 0u;                             // the residual of this block's down columns (wave 0): x is not rewritten before the launch's last instruction
    u32x2v ee = {0u, 0u};                                // [0] the error word, [1] the epoch = this launch's tag (advanced by the fused QKV launch in front: >= 1)
    if constexpr (ATT != 0) {
        // a load the compiler sees (it places the wait at the first use -- the attention role's publishing store, the poll): an asm load would have to be
        // waited for by hand in front of the role's own requests
        unsigned zero_off = 0;
        asm volatile("" : "+v"(zero_off));
        ee = load_granule(reinterpret_cast<const u32x2v*>(p.sync), zero_off);
    } else {
        if (wave == 0) {
            const q4_half* pr = p.xio + c0d + (lane < (unsigned)nc2 ? lane : 0u);
            asm volatile("global_load_ushort %0, %1, off" : "=&v"(resid_raw) : "v"(pr) : "memory");
        }
        const unsigned* pe = p.sync;
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(ee) : "v"(pe) : "memory");
    }
    constexpr int NSIDE = (int)(L::NS_S + L::NS_Z);      // 5 side pieces per gate/up matrix
    auto issue_sides_of = [&](const int wave) {          // the side piece wave `wave` is in charge of (a wave may stand in for another one)
        if (wave < 2 * NSIDE) {
            const int m = wave >= NSIDE, q = wave - NSIDE * m;
            if (q < (int)L::NS_S) {
                const __amdgpu_buffer_rsrc_t rs = rsrc_from(a.m[m].s, c0 * (G * 2u), (unsigned)(a.N * a.sh * 2));
                dma_piece_default(L::SIDE_S + m * L::SIDE_S_BYTES + q * 1024u, voff + (unsigned)q * 1024u, rs, 0u);
            } else {
                const __amdgpu_buffer_rsrc_t rz = rsrc_from(a.m[m].z, c0 * (ZW * 4u), (unsigned)(a.N * a.pzh * 4));
                dma_piece_default(L::SIDE_Z + m * L::SIDE_Z_BYTES + (q - (int)L::NS_S) * 1024u, voff + (unsigned)(q - (int)L::NS_S) * 1024u, rz, 0u);
            }
        } else if (wave < 2 * NSIDE + 4) {                   // the down projection's scales (three pieces) and zeros (one) of this block's columns
            const int q = wave - 2 * NSIDE;
            if (q < 3) {
                const __amdgpu_buffer_rsrc_t rs = rsrc_from(p.d.s, c0d * (unsigned)p.sh * 2u, (unsigned)(p.Nd * p.sh * 2));
                dma_piece_default(P::DSIDE_S + (unsigned)q * 1024u, voff + (unsigned)q * 1024u, rs, 0u);
            } else {
                const __amdgpu_buffer_rsrc_t rz = rsrc_from(p.d.z, c0d * (unsigned)p.pzh * 4u, (unsigned)(p.Nd * p.pzh * 4));
                dma_piece_default(P::DSIDE_Z, voff, rz, 0u);
            }
        }
    };
    auto issue_sides = [&]() { issue_sides_of(wave); };
    if (!ATT) {
        issue_sides();
        block_barrier_lds();  // the x loads are queued on this CU in front of every weight piece (the path returns in order)
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mat ? arg_w1 : arg_w0), 0, (int)wbytes, 0x00020000);
    const unsigned ring = L::RING + (unsigned)wave * (D * 1024u);
    const unsigned soff0 = (c0 + ((unsigned)gw >> 1)) * CB;
    auto issue2 = [&](int i, int ks) {
        const unsigned dst = ring + (unsigned)((TS * i + ks) & (D - 1)) * 1024u, so = soff0 + (unsigned)i * (8u * CB) + (unsigned)ks * 1024u;
        dma_piece(dst, voff, rw, so);
    };
    // the stream's continuation: the down projection's weights of this block -- its output columns are ONE contiguous run of the tensor, which lands in
    // LDS as it lies in memory. A wave requests the pieces of its OWN phase-2 units (column wave / 2 + 8 i, k-part wave % 2: three pieces of 64, 64 and
    // ku - 128 uint4), so that its own vmcnt tells it when they have landed: no block barrier between the stream and the unpacking below.
    const unsigned colbytes2 = (unsigned)p.pw4 * 16u, share = (unsigned)nc2 * colbytes2;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.d.w) + (size_t)c0d * colbytes2), 0, (int)share, 0x00020000);
    const int nu2 = (2 * nc2 - wave + 15) >> 4;                       // this wave's units of phase 2 (<= 2)
    const unsigned ku = (unsigned)p.ku, ubase = (unsigned)(wave & 1) * ku;
    const unsigned uend = ubase + ku < (unsigned)p.pw4 ? ubase + ku : (unsigned)p.pw4;
    const int ndp = 3 * nu2;
    auto issue_down = [&](int r) {                                     // piece r = 3 i + ks
        const unsigned i = r >= 3 ? 1u : 0u, ks = (unsigned)r - 3u * i;
        const unsigned lc = ((unsigned)wave >> 1) + 8u * i;
        const unsigned first = lc * (unsigned)p.pw4 + ubase + 64u * ks;  // in uint4 units from the block's first column
        if (64u * ks + lane < ku && ubase + 64u * ks + lane < uend) dma_piece(P::DW + first * 16u, voff, rd, first * 16u);
    };
    const int nstream = npieces + ndp;
    auto prime_ring = [&]() {
#pragma unroll
        for (int k = 0; k < D; k++)
            if (k < npieces) issue2(k / TS, k % TS);
    };
    if (!ATT) prime_ring();

    u32x4* xs = reinterpret_cast<u32x4*>(smem + L::XS);
    float* sx = reinterpret_cast<float*>(smem + L::SX);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    float* tot = reinterpret_cast<float*>(smem + L::TOT);
    if constexpr (ATT != 0) {
        // ---- phases A and O. Blocks [0, natt) -- half of the launch -- run one (head, V slice) unit of the attention each: attention_oproj_kernel's role
        // (layer_attn.h, forms 5 / 6) on sixteen waves; scores and softmax statistics do not depend on the wave count, the P.V pass stays on eight waves, so every
        // output sums the same terms in the same order. The other half are the output projection: 32 columns a block, two a wave, weights in registers since
        // entry -- gemv_q4_kernel<MODE_PLAIN, 2, 2, false, 5, 1>'s arithmetic (the o-proj role). Either half then has nothing to do until x exists: it requests
        // its side data, its ring and SIX more gate/up pieces a wave (units 1 .. 3) into the region the down weights will take later -- HBM has little else to
        // do in these phases, and 8 of a wave's 10 .. 12 pieces are in LDS when phase 1 begins.
        const bool att_block = blockIdx.x < la.natt;
        const unsigned jb = blockIdx.x - la.natt;              // o-proj block: columns [32 jb, 32 jb + 32)
        // everything wave w wants in LDS when phase 1 begins, requested by this wave (w = wave, or a wave of the same parity -- same matrix, same k-part -- that
        // must not queue requests now): its side piece, its ring's first two pieces, pieces 2 .. 4 ahead, its three pieces of the down columns 8 .. 15
        auto issue_ahead_of = [&](const int w) {
            issue_sides_of(w);
            const unsigned so = (c0 + ((unsigned)w >> 1)) * CB;
#pragma unroll
            for (int j = 0; j < 2 + P::NCREDIT; j++) {
                const unsigned dst = j < 2 ? L::RING + (unsigned)w * (D * 1024u) + (unsigned)j * 1024u : P::credit_slot(16u * (unsigned)(j - 2) + (unsigned)w, colbytes2);
                dma_piece(dst, voff, rw, so + (unsigned)(j / TS) * (8u * CB) + (unsigned)(j % TS) * 1024u);
            }
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                const unsigned first = (((unsigned)w >> 1) + 8u) * (unsigned)p.pw4 + ubase + 64u * (unsigned)ks;
                if (64u * (unsigned)ks + lane < ku && ubase + 64u * (unsigned)ks + lane < uend) dma_piece(P::DW + first * 16u, voff, rd, first * 16u);
            }
        };
        const unsigned tagA = ee[1];            // raw (arithmetic on it here would pull the load's wait in front of the role's requests): tag_add is 0 in this form
        const unsigned nchA = (unsigned)a.K >> 3;                          // 512 chunks of eight halves: one per thread of waves 0 .. 7
        const bool gathA = tid < nchA;
        unsigned deadA = 0u;
        auto gather_vec = [&](const u32x2v* gran, u32x4& out, u32x4& other, const bool lines) {      // other: an asm load of the caller that lands under the pass
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gran, 0, (int)((lines ? line_area_words(nchA * 4u) : gran_area_words(nchA * 4u)) * 4u), 0x00020000);
            bool need = true, failed = false;
            unsigned tries = 0;
            for (;;) {
                u32x4 g0, g1;
                const unsigned o0 = need ? (lines ? line_slot(4u * tid) : gran_slot(4u * tid)) * 8u : 0x7FFFFF00u;
                // (sc1: past the L1; see the third phase's gather)
                asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen sc1\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16 sc1" : "=&v"(g0), "=&v"(g1) : "v"(o0), "s"(rg) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(g0), "+v"(g1), "+v"(other) : : "memory");
                if (need && g0[1] == tagA && g0[3] == tagA && g1[1] == tagA && g1[3] == tagA) { out = (u32x4){g0[0], g0[2], g1[0], g1[2]}; need = false; }
                if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
                if (++tries >= FP_POLL_LIMIT || deadA != 0u) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed && deadA == 0u && lane == 0) __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (att_block) {
            Handoff ho = {};
            ho.tag = tagA;
            ho.pub = la.agran;
#ifdef Q4_PROFILING
            ho.mute = p.mute != 0;
#endif
            constexpr int UA = ATT == 1 ? 2 : 4;
            attention_body<16, UA, STRIP_WAVES, 2, false, 4, ATT == 2 ? 128 : 0, 8>(la.att, (int)(blockIdx.x % la.nheads), ho, (int)(blockIdx.x / la.nheads), P::ATT_LDS);
            if (wave == 0) FPSTAMP2(40);
            block_barrier_lds();                           // the role's scratch lies over the x staging area
            issue_ahead_of(wave);
            deadA = ee[0];
        } else {
            // the o-proj block's requests, all at entry: two columns a wave with their zero words and scales (as gemv_q4.h loads them), the residual
            const unsigned col0 = 32u * jb + 2u * (unsigned)wave;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)la.o.w, 0, a.K * 2048, 0x00020000);
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)la.o.z, 0, a.K * 16, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)la.o.s, 0, a.K * 64, 0x00020000);
            u32x4 ow[2][2];
            unsigned ozw[2][2];
            uint16_t osc[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const unsigned jj = 64u * (unsigned)ks + lane;
                    ozw[c][ks] = __builtin_amdgcn_raw_buffer_load_b32(rz, (jj >> 5) * 4u, (col0 + (unsigned)c) * 16u, 0);
                    osc[c][ks] = __builtin_amdgcn_raw_buffer_load_b16(rs, (jj >> 2) * 2u, (col0 + (unsigned)c) * 64u, 0);
                    ow[c][ks] = __builtin_amdgcn_raw_buffer_load_b128(ro, jj * 16u, (col0 + (unsigned)c) * 2048u, 2);
                }
            unsigned ores = 0u;
            if (tid < 32u) ores = reinterpret_cast<const uint16_t*>(p.xio)[32u * jb + tid];     // the o-proj's residual (:323, accum)
            __builtin_amdgcn_sched_barrier(0);
            // waves 8 .. 15 (they do not gather) request what they want ahead of phase 1 now: landed before the attention output exists. Waves 0 .. 7 wait until
            // the block has published: requests of theirs in front of the gather would delay it (the CU's path returns in order). (All of it by waves 8 .. 15 at
            // entry, for both halves of the block, was measured: the bytes then sit in front of the gather all the same -- 1001 -> 978 tokens/s.)
            if (wave >= NSTAGE) issue_ahead_of(wave);
            if (wave == 0) FPSTAMP2(40);
            deadA = ee[0];
            // ---- seam A -> O: one lane polls ONE granule (the last of head jb % heads: few pollers per line), then waves 0 .. 7 read the vector, every granule
            // validating itself (layer_attn.h's protocol)
            if (tid == 0 && deadA == 0u) {
                const unsigned sentinel = (jb % la.nheads) * ((unsigned)la.att.head_size >> 1) + ((unsigned)la.att.head_size >> 1) - 1u;
                unsigned i = 0;
                while (load_granule(la.agran, line_slot(sentinel))[1] != tagA && ++i < POLL_LIMIT) __builtin_amdgcn_s_sleep(2);
                if (i >= POLL_LIMIT) __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            block_barrier_lds();
            if (gathA) {
                u32x4 av = {0u, 0u, 0u, 0u};
                gather_vec(la.agran, av, wraw, true);       // (the attention output: whole lines, line_slot)
                // stage: gemv_q4_body's layout for a vector that is multiplied as it is (odd units negated)
                const unsigned sgn = q4_stage_sign_bits(tid);
                const u32x4 pv = permute_x8(q4_signed_x(av, sgn));
                const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
                float cb = 0.f;
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
                cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);
                const unsigned j = tid >> 2, d = tid & 3u;
                xs[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
                if (d == 0) sx[j] = cb * -9.5367431640625e-07f;
            }
            block_barrier_lds();
            if (wave == 0) FPSTAMP2(41);
            float cso[2] = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                u32x4 X[4];
#pragma unroll
                for (int d = 0; d < 4; d++) X[d] = xs[((ks * 4 + d) << 6) + lane];
                const float corr = sx[ks * 64 + lane];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const u32x4 w = ow[c][ks];
                    float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const unsigned ww = w[d], tt = ww >> 8;
                        acc_e = __builtin_amdgcn_fdot2(as_h2(ww & 0x000F000Fu), as_h2(X[d][0]), acc_e, false);
                        acc_o = __builtin_amdgcn_fdot2(as_h2(ww & 0x00F000F0u), as_h2(X[d][1]), acc_o, false);
                        acc_e = __builtin_amdgcn_fdot2(as_h2(tt & 0x000F000Fu), as_h2(X[d][2]), acc_e, false);
                        acc_o = __builtin_amdgcn_fdot2(as_h2(tt & 0x00F000F0u), as_h2(X[d][3]), acc_o, false);
                    }
                    const float zf = (float)((ozw[c][ks] >> (((lane >> 2) & 7u) * 4u)) & 0xFu);
                    float t = __builtin_fmaf(acc_e, 16.f, acc_o);
                    t = __builtin_fmaf(zf, corr, t);
                    cso[c] = __builtin_fmaf(h2f(osc[c][ks]), t, cso[c]);
                }
            }
            const float total = reduce4_q4(cso[0], cso[1], 0.f, 0.f) * 1048576.f;      // rows 0 and 1 hold the two columns
            float* totO = reinterpret_cast<float*>(smem + P::TOT2);
            if ((lane & 15u) == 0u && (lane >> 4) < 2u) totO[2 * wave + (int)(lane >> 4)] = total;
            block_barrier_lds();
            if (wave == 0) {
                uint16_t xo = 0;
                if (lane < 32u) {
                    float r = totO[lane];
                    r += h2f((uint16_t)ores);                   // gpu_kernels.h:229-230
                    xo = f2h(r);                                // :231 -- as granules only: RunState::x is written once per launch, by the block that owns the
                }                                               // column in phase 2 (two CUs' plain stores to one line may be written back in either order)
                const unsigned partner = (unsigned)__shfl_down((int)xo, 1);
                if ((lane & 1u) == 0u && lane < 32u) store_granule(la.xogran + gran_slot((32u * jb + lane) >> 1), (unsigned)xo | (partner << 16), tagA);
                FPSTAMP2(42);
            }
            if (wave < NSTAGE) issue_ahead_of(wave);
        }
        // ---- seam O -> 1: the residual stream from every o-proj block, the x chain of the FFN half (ffn_strip_kernel's: 512 chunk partials, canonical reduction)
        if (gathA) {
            const u32x4* pw = arg_rms + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wraw) : "v"(pw) : "memory");
            gather_vec(la.xogran, xraw, wraw, false);       // (the residual stream: half-line records, gran_slot)
            // the residual of the block's sixteen down columns (chunks 2 b and 2 b + 1): kept in LDS until phase 2 ends
            if ((tid >> 1) == blockIdx.x) *reinterpret_cast<u32x4*>(smem + P::TOT2 + 128u + (tid & 1u) * 16u) = xraw;
        }
        wait_vmcnt<0>();                                   // the side data, the pieces ahead and the ring's first pieces have landed (phase 1 counts from here)
        if (wave == 1) FPSTAMP2(43);
    } else {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xraw), "+v"(wraw), "+v"(ee), "+v"(resid_raw) : "n"(D) : "memory");   // all but the weight pieces (every wave has units: ffn_pair_covers)
    }

    // ---- x chain (ffn_strip_kernel's)
    if (wave == 0) FPSTAMP(1);
    if (NORM) {
        if (tid < (unsigned)(TS * 256)) part[tid] = stager ? sumsq8(xraw, 0.f) : 0.f;
        block_barrier_lds();
        if (wave == 0) FPSTAMP(2);
    }
    if (stager) {
        float ss = 1.f;
        if (NORM) ss = rms_scale_from_partials<TS * 256>(part, TS * 256, a.K);
        u32x4 v = xraw;
        const unsigned sgn = q4_stage_sign_bits(tid);
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
    if (wave == 0) FPSTAMP(3);
    {
        u32x4 X[TS][4];
        float corr[TS];
#pragma unroll
        for (int ks = 0; ks < TS; ks++) {
#pragma unroll
            for (int d = 0; d < 4; d++) X[ks][d] = xs[((ks * 4 + d) << 6) + lane];
            corr[ks] = sx[ks * 64 + lane];
        }
        const unsigned char* wbase = smem + ring + lane * 16u;
        const unsigned char* sbase = smem + L::SIDE_S + mat * L::SIDE_S_BYTES + ((unsigned)(gw >> 1) * G + (lane >> 2)) * 2u;
        const unsigned char* zbase = smem + L::SIDE_Z + mat * L::SIDE_Z_BYTES + ((unsigned)(gw >> 1) * ZW + (lane >> 5)) * 4u;
        const unsigned zsh = ((lane >> 2) & 7u) * 4u;

        // Whole-layer form. Pieces 0, 1 wait in the ring, 2 .. 4 in the region ahead, the rest stream through the ring again (piece j >= 5 in entry (j + 1) % 2: the
        // entry pieces 0 and 1 free). The down columns 8 .. 15 are in LDS already; 0 .. 7 go out as soon as every wave has left the region ahead (a barrier
        // behind piece 4). Requests of a wave in phase 1, in order: G5 G6 | d0 d1 | G7 d2 | G8 | G9 | G10 ...; the waits below count the younger ones.
        auto issue2s = [&](const int j, const int entry) {
            dma_piece(ring + (unsigned)entry * 1024u, voff, rw, soff0 + (unsigned)(j / TS) * (8u * CB) + (unsigned)(j % TS) * 1024u);
        };
        auto group = [&](auto first_c, const int g4) {
            constexpr bool CRED = ATT != 0 && decltype(first_c)::value;
            float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = g4 * 4 + r;
                if (i < nu) {
                    float c = 0.f;
#pragma unroll
                    for (int ks = 0; ks < TS; ks++) {
                        const int j = TS * i + ks;
                        const int e = (TS * r + ks) & (D - 1);
                        const unsigned char* wsrc = wbase + e * 1024;
                        const int jj = TS * r + ks;         // a constant once unrolled
                        if constexpr (CRED) {
                            if (jj >= 2 && jj <= 4) wsrc = smem + P::credit_slot(16u * (unsigned)(jj - 2) + (unsigned)wave, colbytes2) + lane * 16u;   // (landed before phase 1 began)
                            else if (jj >= 5) wsrc = wbase + ((jj + 1) & 1) * 1024;
                            if (jj == 5) wait_vmcnt<3>(); else if (jj == 6) wait_vmcnt<4>(); else if (jj == 7) wait_vmcnt<2>();
                        } else if constexpr (ATT != 0) {
                            wsrc = wbase + ((jj + 1) & 1) * 1024;
                            if (j + 1 < npieces) wait_vmcnt<1>(); else wait_vmcnt<0>();
                        } else {
                            if (j + 1 < nstream) wait_vmcnt<D - 1>(); else wait_vmcnt<0>();     // piece j has landed
                        }
                        const u32x4 w = *reinterpret_cast<const u32x4*>(wsrc);
                        const uint16_t sc = *reinterpret_cast<const uint16_t*>(sbase + (unsigned)i * (8u * G * 2u) + ks * 32);
                        const unsigned zw = *reinterpret_cast<const unsigned*>(zbase + (unsigned)i * (8u * ZW * 4u) + ks * 8);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done: the entry may be refilled
                        if constexpr (CRED) {
                            if (jj < 2) issue2s(jj + 5, jj);
                            if (jj == 4) {
                                block_barrier_lds();       // every wave has left the region ahead
                                if (0 < ndp) issue_down(0);
                                if (1 < ndp) issue_down(1);
                            }
                            if (jj == 5) { issue2s(7, 0); if (2 < ndp) issue_down(2); }
                            if (jj == 6) issue2s(8, 1);
                            if (jj == 7) issue2s(9, 0);
                        } else if constexpr (ATT != 0) {
                            if (j + 2 < npieces) issue2s(j + 2, (jj + 1) & 1);
                        } else {
                            if (j + D < npieces) issue2(i + (ks + D) / TS, (ks + D) % TS);
                            else if (j + D < nstream) issue_down(j + D - npieces);                // ... and behind the last gate/up piece the stream goes on
                        }
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
                        c = __builtin_fmaf(h2f(sc), t, c);
                    }
                    cs[r] = c;
                }
            }
            const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;
            const int row = lane >> 4;
            if ((lane & 15u) == 0 && g4 * 4 + row < nu) tot[gw + 16 * (g4 * 4 + row)] = total;   // [column][matrix] = unit index
        };
        if constexpr (ATT != 0) {
            group(std::true_type{}, 0);
            for (int g4 = 1; g4 * 4 < nu; g4++) group(std::false_type{}, g4);
        } else {
            for (int g4 = 0; g4 * 4 < nu; g4++) group(std::false_type{}, g4);
        }
    }
    FPSTAMP(32 + wave);
    int rnext = ndp < D ? ndp : D;                     // down pieces this wave has requested so far
    if (ATT != 0) rnext = ndp;                         // (whole-layer form: all of them)
    block_barrier_lds();                               // barrier A: the block's gate/up totals are in LDS, its rings are dead

    // ---- the seam
    const unsigned tag = ee[1] + p.tag_add, dead = ee[0];
    u32x4 ha = {0u, 0u, 0u, 0u}, hb2 = {0u, 0u, 0u, 0u};    // this thread's chunks of hb (8 halves each)
    const unsigned gt = tid - 64u;                          // gather thread (waves 1 .. 15): chunks gt and gt + 960
    const unsigned nch2 = (unsigned)p.Kd >> 3;
    const unsigned tail = ku - 128u;
    // phase 2's weights, unpacked while the wave waits for the vector: [unit][slot][dword] x {n0 n4, n1 n5 << 4, n2 n6, n3 n7 << 4} as fp16 denormal pairs
    // (the last piece of each phase stays packed -- four registers instead of sixteen: the kernel lives at the 128 registers a lane has at sixteen waves per CU)
    unsigned pm[2][3][16];
    u32x4 wlast = {0u, 0u, 0u, 0u};
    auto unpack = [&](const int ifirst, const int ilast) {
#pragma unroll
        for (int i = ifirst; i <= ilast; i++) {
            const unsigned lc = ((unsigned)wave >> 1) + 8u * (unsigned)i;
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                const unsigned unit = ubase + 64u * (unsigned)ks + lane;
                const unsigned uj = unit < uend ? unit : uend - 1u;
                u32x4 w = {0u, 0u, 0u, 0u};
                if (i < nu2) w = *reinterpret_cast<const u32x4*>(smem + P::DW + (lc * (unsigned)p.pw4 + uj) * 16u);
                if (i == 1 && ks == 2) { wlast = w; asm volatile("" : "+v"(wlast)); continue; }
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned ww = w[d], tt = ww >> 8;
                    pm[i][ks][4 * d + 0] = ww & 0x000F000Fu;
                    pm[i][ks][4 * d + 1] = ww & 0x00F000F0u;
                    pm[i][ks][4 * d + 2] = tt & 0x000F000Fu;
                    pm[i][ks][4 * d + 3] = tt & 0x00F000F0u;
                }
                // the masks are computed HERE, while the wave waits for the vector: left alone hipcc sinks them behind barrier B, next to the dot products (seen in the ISA)
                asm volatile("" : "+v"(pm[i][ks][0]), "+v"(pm[i][ks][1]), "+v"(pm[i][ks][2]), "+v"(pm[i][ks][3]), "+v"(pm[i][ks][4]), "+v"(pm[i][ks][5]), "+v"(pm[i][ks][6]), "+v"(pm[i][ks][7]),
                             "+v"(pm[i][ks][8]), "+v"(pm[i][ks][9]), "+v"(pm[i][ks][10]), "+v"(pm[i][ks][11]), "+v"(pm[i][ks][12]), "+v"(pm[i][ks][13]), "+v"(pm[i][ks][14]), "+v"(pm[i][ks][15]));
            }
        }
    };
    if (wave == 0) {
        FPSTAMP(4);
        if ((int)lane * 2 < nc) {      // SiLU(gate) * up of one column pair (gpu_kernels.h:271-272), the epilogue of ffn_strip_kernel
            float v0 = tot[4 * lane], v1 = tot[4 * lane + 2];
            const float u0 = tot[4 * lane + 1], u1 = tot[4 * lane + 3];
            v0 *= 1.0f / (1.0f + expf(-v0)); v0 *= u0;
            v1 *= 1.0f / (1.0f + expf(-v1)); v1 *= u1;
            const unsigned pk = (unsigned)f2h(v0) | ((unsigned)f2h(v1) << 16);
            *reinterpret_cast<unsigned*>(a.out[0] + c0 + 2u * lane) = pk;          // RunState::hb, as the launch sequence leaves it
#ifdef Q4_PROFILING
            if (!p.mute)
#endif
            store_granule(p.gran + bp + lane, pk, tag);
        }
        FPSTAMP(5);
        // (A sentinel poll by this wave -- the last granule of every block's slice, sc1, an LDS flag for the gathering waves when all carry the tag -- was
        // measured: 256 CUs polling 256 lines slow the down stream itself by 1 us, 976 -> 952 tokens/s. The gathering waves time their first pass by their own
        // pieces' landing instead: FfnPairArgs::pre.)
        while (rnext < ndp) issue_down(rnext++);
        FPSTAMP(16);
        wait_vmcnt<0>();
        FPSTAMP(48);
        unpack(0, 1);
    } else {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.gran, 0, (int)(nch2 * 32u), 0x00020000);
        const unsigned u0 = gt, u1 = gt + 960u;
        bool need0 = u0 < nch2, need1 = u1 < nch2;
        while (rnext < ndp) issue_down(rnext++);           // the rest of this wave's share of the down stream
        if (p.pre & 2u) wait_vmcnt_upto15((int)((p.pre >> 4) & 15u));   // (mode bit 1: the first pass leaves when all but `pre >> 4` of the wave's pieces have landed)
        if (STAMPS && wave == 1) FPSTAMP(8);
        unsigned tries = 0, passes = 0;
        bool failed = false;
        for (;;) {
            u32x4 g0, g1, g2, g3;
            // A chunk that is complete is not read again: its lanes aim past the descriptor's range, which returns zeros without a memory request (no
            // branch: no copies of a destination register before its data has landed). asm: hipcc must not count these loads, it cannot see the DMA pieces.
            const unsigned o0 = need0 ? u0 * 32u : 0x7FFFFF00u, o1 = need1 ? u1 * 32u : 0x7FFFFF00u;
            if (passes == 0 && (p.pre & 1u))      // mode bit 0: the first pass with plain loads -- the L1 is cold for these lines, the XCD's L2 serves the neighbours
                asm volatile("buffer_load_dwordx4 %0, %4, %6, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen offset:16\n\t"
                             "buffer_load_dwordx4 %2, %5, %6, 0 offen\n\tbuffer_load_dwordx4 %3, %5, %6, 0 offen offset:16"
                             : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(o0), "v"(o1), "s"(rg) : "memory");
            else                                   // sc1: past the L1 (a line this CU read too early sits there stale)
                asm volatile("buffer_load_dwordx4 %0, %4, %6, 0 offen sc1\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen offset:16 sc1\n\t"
                             "buffer_load_dwordx4 %2, %5, %6, 0 offen sc1\n\tbuffer_load_dwordx4 %3, %5, %6, 0 offen offset:16 sc1"
                             : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(o0), "v"(o1), "s"(rg) : "memory");
            if (passes == 0) {                     // under the first pass: the wave's own pieces (requested in front of it: they return first) are unpacked
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                FPSTAMP(48 + wave);
                unpack(0, 0);                      // (the second unit behind the pass, when its destination registers are free again)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : : "memory");   // ONE statement names the destinations, behind the wait
            if (need0 && g0[1] == tag && g0[3] == tag && g1[1] == tag && g1[3] == tag) { ha = (u32x4){g0[0], g0[2], g1[0], g1[2]}; need0 = false; }
            if (need1 && g2[1] == tag && g2[3] == tag && g3[1] == tag && g3[3] == tag) { hb2 = (u32x4){g2[0], g2[2], g3[0], g3[2]}; need1 = false; }
            const bool pending = __builtin_amdgcn_ballot_w64(need0 || need1) != 0ull;
            if (STAMPS && passes == 0 && wave == 1) FPSTAMP(6);
            passes++;
            if (!pending) break;
            if (++tries >= FP_POLL_LIMIT || dead != 0u) { failed = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        if (failed && dead == 0u && lane == 0) __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unpack(1, 1);
        if (STAMPS) { FPSTAMP(16 + wave); if (wave == 1) { FPSTAMP(7); if (lane == 0) st[12] = passes; } }
        // stage: down_strip_kernel's (gemv_q4_body's) permuted layout, odd units negated
        u32x4* xs2 = reinterpret_cast<u32x4*>(smem + P::XS2);
        float* sx2 = reinterpret_cast<float*>(smem + P::SX2);
        const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
        const unsigned sgn = q4_stage_sign_bits(tid);      // (tid and gt, gt and gt + 960: the same unit parity)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const unsigned u = i ? u1 : u0;
            u32x4 v = i ? hb2 : ha;
            v = q4_signed_x(v, sgn);
            const u32x4 pv = permute_x8(v);
            float cb = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
            cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);
            const unsigned j = u >> 2, d = u & 3u;
            if (u < nch2) {
                xs2[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
                if (d == 0) sx2[j] = cb * -9.5367431640625e-07f;
            }
        }
    }
    block_barrier_lds();                               // barrier B: hb staged (every wave's own down pieces have landed and are unpacked)
    if (wave == 1) FPSTAMP(9);
    // ---- phase 3's geometry: the block owns the RoPE pairs pair0 .. pair0 + 7 of q, k and v: 48 columns cw (matrix cw / 16, RoPE half (cw % 16) / 8, pair cw % 8),
    // wave w multiplies cw = 3 w + j
    const unsigned hp3 = QKV ? (unsigned)q3.head_size >> 1 : 1u;
    const unsigned pair0 = QKV ? blockIdx.x * q3.ppb : 0u;             // the block's first RoPE pair (of every matrix); its pairs stay inside one head
    const unsigned n0q = QKV ? (pair0 / hp3) * (unsigned)q3.head_size + pair0 % hp3 : 0u;   // first column of the block's lower run; the upper run: + head_size / 2
    // Its stream goes out PIECE BY PIECE during phase 2, one request in front of each of the wave's six dot-product steps (the down weights sit in registers
    // since barrier B: their LDS is free). All at once -- at barrier B, or behind the dot products -- the 100 requests of a CU stall in ISSUE (a CU takes
    // 32-40 KiB in flight) and hold their waves: phase 2 then takes 3.6 us instead of 1.9, or barrier C is reached 2.2 us late (tools/lab/ffn_pair_check.py).
    auto issue_qkv = [&](const int r) {                 // request r = 0 .. 5 of this wave: column cw = 3 wave + r / 2, k-slot r % 2; with r = 0 also a piece of side data
        const unsigned cw = 3u * (unsigned)wave + (unsigned)(r >> 1), m = cw >> 4, n = n0q + (((cw >> 3) & 1u) ? hp3 : 0u) + (cw & 7u);
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)q3.m[m].w, 0, (int)(a.K * 2048), 0x00020000);   // dim columns of 2 KiB
        dma_piece_free(P::DW + cw * 2048u + (unsigned)(r & 1) * 1024u, voff, rq, n * 2048u + (unsigned)(r & 1) * 1024u);
        if (r == 0 && wave < 12) {                     // the six runs of eight columns: 512 B of scales (waves 0 .. 5), 128 B of zero words (6 .. 11)
            const unsigned run = wave < 6 ? (unsigned)wave : (unsigned)wave - 6u, ms = run >> 1, ns = n0q + ((run & 1u) ? hp3 : 0u);
            if (wave < 6) {
                const __amdgpu_buffer_rsrc_t rs = rsrc_from(q3.m[ms].s, ns * 64u, (unsigned)a.K * 64u);
                if (lane < 32u) dma_piece_default_free(P::QSIDE_S + run * 512u, voff, rs, 0u);
            } else {
                const __amdgpu_buffer_rsrc_t rz = rsrc_from(q3.m[ms].z, ns * 16u, (unsigned)a.K * 16u);
                if (lane < 8u) dma_piece_default_free(P::QSIDE_Z + run * 128u, voff, rz, 0u);
            }
        }
    };
    // ---- phase 2: down_strip_kernel<3, false>'s units; a k-slot's inputs are read once for both units of the wave
    const uint16_t resid = (uint16_t)resid_raw;
    {
        const u32x4* xs2 = reinterpret_cast<const u32x4*>(smem + P::XS2);
        const float* sx2 = reinterpret_cast<const float*>(smem + P::SX2);
        float* tot2 = reinterpret_cast<float*>(smem + P::TOT2);
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef Q4_PROFILING
        if (!(p.pre & 8u))      // (ablation, profiling build: phase 2 without its dot products -- what they cost the token; results are garbage)
#endif
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            const bool last = ks == 2;
            const unsigned unit = ubase + 64u * (unsigned)ks + lane;
            const bool live = !last || (lane < tail && unit < uend);
            const unsigned uj = unit < uend ? unit : uend - 1u;
            const unsigned xrow = ((uj >> 6) << 8) + (uj & 63u);
            u32x4 X[4];
#pragma unroll
            for (int d = 0; d < 4; d++) X[d] = xs2[xrow + (d << 6)];
            const float corr = sx2[uj];
            const unsigned zsh = ((uj >> 2) & 7u) * 4u;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (QKV) issue_qkv(2 * ks + i);                 // (three requests under the dot products and three behind them: 1009 -> 1000 tokens/s; 72 KiB of the
                                                                //  CU's 96 under them, 24 behind: 1012 -> 1005)
                if (i < nu2) {
                    const unsigned lc = ((unsigned)wave >> 1) + 8u * (unsigned)i;
                    const uint16_t sc = *reinterpret_cast<const uint16_t*>(smem + P::DSIDE_S + (lc * (unsigned)p.sh + (uj >> 2)) * 2u);
                    const unsigned zw = *reinterpret_cast<const unsigned*>(smem + P::DSIDE_Z + (lc * (unsigned)p.pzh + (uj >> 5)) * 4u);
                    float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        unsigned m0, m1, m2, m3;
                        if (i == 1 && ks == 2) { const unsigned ww = wlast[d], tt = ww >> 8; m0 = ww & 0x000F000Fu; m1 = ww & 0x00F000F0u; m2 = tt & 0x000F000Fu; m3 = tt & 0x00F000F0u; }
                        else { m0 = pm[i][ks][4 * d + 0]; m1 = pm[i][ks][4 * d + 1]; m2 = pm[i][ks][4 * d + 2]; m3 = pm[i][ks][4 * d + 3]; }
                        acc_e = __builtin_amdgcn_fdot2(as_h2(m0), as_h2(X[d][0]), acc_e, false);
                        acc_o = __builtin_amdgcn_fdot2(as_h2(m1), as_h2(X[d][1]), acc_o, false);
                        acc_e = __builtin_amdgcn_fdot2(as_h2(m2), as_h2(X[d][2]), acc_e, false);
                        acc_o = __builtin_amdgcn_fdot2(as_h2(m3), as_h2(X[d][3]), acc_o, false);
                    }
                    const float zf = (float)((zw >> zsh) & 0xFu);
                    float t = __builtin_fmaf(acc_e, 16.f, acc_o);
                    t = __builtin_fmaf(zf, corr, t);
                    if (last) {         // an ordinary last slot: its lanes past the part's end multiply zero inputs in gemv_q4.h, fma(s, 0, c) = c
                        const float f = __builtin_fmaf(h2f(sc), t, cs[i]);
                        cs[i] = live ? f : cs[i];
                    } else {
                        cs[i] = __builtin_fmaf(h2f(sc), t, cs[i]);
                    }
                }
            }
        }
        const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;
        const int row = lane >> 4;
        if ((lane & 15u) == 0 && row < nu2) tot2[wave + 16 * row] = total;      // [column][k-part] = unit index
        if (wave == 1) FPSTAMP(10);
        FPSTAMP2(wave);
        block_barrier_lds();
        if (wave == 0) FPSTAMP2(32);
        uint16_t xnew = 0;
        if ((int)tid < nc2) {
            float r = tot2[2 * tid];
            r += tot2[2 * tid + 1];                     // fixed order: k-parts from the lowest up
            r += h2f(ATT != 0 ? reinterpret_cast<const uint16_t*>(smem + P::TOT2 + 128u)[tid] : resid);   // gpu_kernels.h:229-230
            xnew = f2h(r);
            p.xio[c0d + tid] = xnew;                    // :231
        }
        if (QKV && wave == 0) {                         // the block's slice of the new residual stream, for every other block: column pairs (c0d and nc2 are even)
            const unsigned partner = (unsigned)__shfl_down((int)xnew, 1);
            if ((lane & 1u) == 0u && (int)lane < nc2) store_granule(q3.xgran + gran_slot((c0d + lane) >> 1), (unsigned)xnew | (partner << 16), tag);
        }
    }
    if (QKV) {
        if (wave == 0) { FPSTAMP(13); FPSTAMP2(33); }
        // ---- the second seam: x from every block (dim / 8 chunks of eight halves: one per thread of waves 0 .. 7), the next layer's norm weights, the
        // position and its (cos, sin) entries; meanwhile every wave unpacks its q / k / v pieces
        const unsigned nch3 = (unsigned)a.K >> 3;                        // 512
        const bool gath = tid < nch3;
        u32x4 xr3 = {0u, 0u, 0u, 0u}, wr3 = {0u, 0u, 0u, 0u};
        unsigned pm3[3][2][16];
        u32x4 wlast3 = {0u, 0u, 0u, 0u};
        auto unpack3 = [&](const int jfirst, const int jlast) {
#pragma unroll
            for (int j = jfirst; j <= jlast; j++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const u32x4 w = *reinterpret_cast<const u32x4*>(smem + P::DW + ((3u * (unsigned)wave + (unsigned)j) * 128u + 64u * (unsigned)ks + lane) * 16u);
                    if (j == 2 && ks == 1) { wlast3 = w; asm volatile("" : "+v"(wlast3)); continue; }
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const unsigned ww = w[d], tt = ww >> 8;
                        pm3[j][ks][4 * d + 0] = ww & 0x000F000Fu;
                        pm3[j][ks][4 * d + 1] = ww & 0x00F000F0u;
                        pm3[j][ks][4 * d + 2] = tt & 0x000F000Fu;
                        pm3[j][ks][4 * d + 3] = tt & 0x00F000F0u;
                    }
                    asm volatile("" : "+v"(pm3[j][ks][0]), "+v"(pm3[j][ks][1]), "+v"(pm3[j][ks][2]), "+v"(pm3[j][ks][3]), "+v"(pm3[j][ks][4]), "+v"(pm3[j][ks][5]), "+v"(pm3[j][ks][6]), "+v"(pm3[j][ks][7]),
                                 "+v"(pm3[j][ks][8]), "+v"(pm3[j][ks][9]), "+v"(pm3[j][ks][10]), "+v"(pm3[j][ks][11]), "+v"(pm3[j][ks][12]), "+v"(pm3[j][ks][13]), "+v"(pm3[j][ks][14]), "+v"(pm3[j][ks][15]));
                }
        };
        wait_vmcnt<0>();                                 // this wave's q / k / v pieces have landed (wave 0: its stores are acknowledged)
        if (gath) {
            const u32x4* pw = reinterpret_cast<const u32x4*>(q3.rms_w) + tid;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wr3) : "v"(pw) : "memory");
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)q3.xgran, 0, (int)(gran_area_words(nch3 * 4u) * 4u), 0x00020000);
            const unsigned xoff3 = gran_slot(4u * tid) * 8u;      // chunk tid = granules 4 tid .. 4 tid + 3, inside one record
            bool need = true;
            unsigned tries = 0, passes = 0;
            bool failed = false;
            for (;;) {
                u32x4 g0, g1;
                const unsigned o0 = need ? xoff3 : 0x7FFFFF00u;
                // sc1 from the first pass on: the blocks reach this seam up to a microsecond apart, an early plain read would leave stale lines in this CU's L1 (the
                // XCDs' L2s are kept coherent, tools/lab/t_l2stale.hip). Plain loads with the L1 invalidated in front of every retry were measured: `buffer_inv sc1`
                // takes ~10 us a pass (1019 -> 728 tokens/s), `buffer_inv sc0` does not invalidate the L1 at all (the waits run out)
                asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen sc1\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16 sc1" : "=&v"(g0), "=&v"(g1) : "v"(o0), "s"(rg) : "memory");
                if (passes == 0) unpack3(0, 1);         // under the first pass (the third column behind it, when the pass's registers are free again)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(g0), "+v"(g1), "+v"(wr3) : : "memory");
                if (need && g0[1] == tag && g0[3] == tag && g1[1] == tag && g1[3] == tag) { xr3 = (u32x4){g0[0], g0[2], g1[0], g1[2]}; need = false; }
                if (STAMPS && passes == 0 && wave == 1) FPSTAMP2(34);
                passes++;
                if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
                if (++tries >= FP_POLL_LIMIT || dead != 0u) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed && dead == 0u && lane == 0) __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (STAMPS && wave == 1) { FPSTAMP2(35); if (lane == 0) reinterpret_cast<unsigned long long*>(smem + P::STAMP2)[39] = passes; }
            unpack3(2, 2);
        } else {
            unpack3(0, 2);
        }
        if (wave == 1) FPSTAMP(14);
        // the x chain of gemv_q4_kernel<MODE_QKV, 2, 4, true>: 512 chunk partials, the canonical reduction, the norm folded into the staging
        u32x4* xs3 = reinterpret_cast<u32x4*>(smem + L::XS);
        float* sx3 = reinterpret_cast<float*>(smem + L::SX);
        float* part3 = reinterpret_cast<float*>(smem + L::PART);
        float* tot3 = reinterpret_cast<float*>(smem + L::TOT);
        if (gath) part3[tid] = sumsq8(xr3, 0.f);
        block_barrier_lds();
        if (wave == 1) FPSTAMP2(36);
        if (gath) {
            const float ss = rms_scale_from_partials<512>(part3, 512, a.K);
            const unsigned sgn = q4_stage_sign_bits(tid);
            const u32x4 v = rms_apply8(xr3, wr3, q4_signed_scale(ss, sgn));
            const u32x4 pv = permute_x8(v);
            const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
            float cb = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
            cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);
            const unsigned j = tid >> 2, d = tid & 3u;
            xs3[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
            if (d == 0) sx3[j] = cb * -9.5367431640625e-07f;
        }
        // position and rotation of the block's pairs (threads 0 .. 47: one column each), requested now, used behind the dot products
        int pos3 = 0;
        float2 cs3 = {1.f, 0.f};
        if (tid < 48u) {
            pos3 = *q3.pPos;
            cs3 = q3.rope_table[(size_t)pos3 * hp3 + (pair0 % hp3) + (tid & 7u)];
        }
        block_barrier_lds();                             // x staged
        if (wave == 1) { FPSTAMP(15); FPSTAMP2(37); }
        {
            float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                u32x4 X[4];
#pragma unroll
                for (int d = 0; d < 4; d++) X[d] = xs3[((ks * 4 + d) << 6) + lane];
                const float corr = sx3[ks * 64 + lane];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const unsigned cw = 3u * (unsigned)wave + (unsigned)j;
                    const uint16_t sc = *reinterpret_cast<const uint16_t*>(smem + P::QSIDE_S + (cw * 32u + 16u * (unsigned)ks + (lane >> 2)) * 2u);
                    const unsigned zw = *reinterpret_cast<const unsigned*>(smem + P::QSIDE_Z + (cw * 4u + 2u * (unsigned)ks + (lane >> 5)) * 4u);
                    float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        unsigned m0, m1, m2, m3;
                        if (j == 2 && ks == 1) { const unsigned ww = wlast3[d], tt = ww >> 8; m0 = ww & 0x000F000Fu; m1 = ww & 0x00F000F0u; m2 = tt & 0x000F000Fu; m3 = tt & 0x00F000F0u; }
                        else { m0 = pm3[j][ks][4 * d + 0]; m1 = pm3[j][ks][4 * d + 1]; m2 = pm3[j][ks][4 * d + 2]; m3 = pm3[j][ks][4 * d + 3]; }
                        acc_e = __builtin_amdgcn_fdot2(as_h2(m0), as_h2(X[d][0]), acc_e, false);
                        acc_o = __builtin_amdgcn_fdot2(as_h2(m1), as_h2(X[d][1]), acc_o, false);
                        acc_e = __builtin_amdgcn_fdot2(as_h2(m2), as_h2(X[d][2]), acc_e, false);
                        acc_o = __builtin_amdgcn_fdot2(as_h2(m3), as_h2(X[d][3]), acc_o, false);
                    }
                    const float zf = (float)((zw >> (((lane >> 2) & 7u) * 4u)) & 0xFu);
                    float t = __builtin_fmaf(acc_e, 16.f, acc_o);
                    t = __builtin_fmaf(zf, corr, t);
                    cs[j] = __builtin_fmaf(h2f(sc), t, cs[j]);
                }
            }
            const float total = reduce4_q4(cs[0], cs[1], cs[2], cs[3]) * 1048576.f;
            const int row = lane >> 4;
            if ((lane & 15u) == 0 && row < 3) tot3[3 * wave + row] = total;
        }
        FPSTAMP2(16 + wave);
        block_barrier_lds();
        if (tid < 48u) {                                 // the epilogue of gemv_q4_kernel<MODE_QKV>: RoPERotation_kernel (gpu_kernels.h:332-355) on the fp16-rounded outputs, KV write (:251, 253)
            const unsigned m = tid >> 4, upper = (tid >> 3) & 1u, c = tid & 7u;
            const float mine = tot3[tid];
            float r = mine;
            if (m < 2u) {
                const float other = round_h(tot3[tid ^ 8u]), me = round_h(mine);
                const float fcr = cs3.x, fci = cs3.y;
                r = upper ? (other * fci + me * fcr) : (me * fcr - other * fci);    // :345-346
            }
            const unsigned n = n0q + (upper ? hp3 : 0u) + c;
            q4_half* out = m == 0u ? q3.q : (m == 1u ? q3.kc : q3.vc) + (size_t)pos3 * (size_t)q3.kv_dim;
            out[n] = f2h(r);
        }
        if (wave == 0) FPSTAMP2(38);
        // the epoch of the attention -> o-proj launch behind this one (gemv_q4.h, `bump`): advanced by one thread of the launch, at its very end
        if (q3.bump != nullptr && blockIdx.x == 0 && tid == 0) __hip_atomic_fetch_add(q3.bump, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (STAMPS) {
        if (wave == 0) FPSTAMP(11);
        block_barrier_lds();
        if (p.dbg && tid < 64u) p.dbg[(size_t)blockIdx.x * 64 + tid] = st[tid];
        if (p.dbg && tid < 64u) p.dbg[(size_t)gridDim.x * 64 + (size_t)blockIdx.x * 64 + tid] = reinterpret_cast<unsigned long long*>(smem + P::STAMP2)[tid];
    }
}
#undef FPSTAMP
#undef FPSTAMP2

// Shapes: gate/up K = dim = 4096 (two 1 KiB pieces per column), whole column pairs per CU, 8 .. 28 of them; down projection K = hidden in two k-parts of
// three slots with an ordinary last one (Llama-2-7B: 172 = 64 + 64 + 44 units), at most 16 output columns per CU whose weights fit the 88 KiB the launch keeps
// for them; every CU of the device takes one block, so the stream must not be masked. What decides whether the form RUNS is q4_runtime.hip (fusion level 4).
static bool ffn_pair_shape(const GemvArgs& a, int hidden_k, int dim_n) {
    const int nb = cu_count();
    if (!(a.K == 4096 && a.pw4 == 128 && a.sh == 32 && a.pzh == 4) || a.N != hidden_k || (a.N & 1) || stream_cu_count() != nb || g_ablate != 0) return false;
    const int pairs = a.N / 2;
    if (pairs / nb < 8 || 2 * divUp(pairs, nb) > STRIP_NCMAX) return false;
    const QGeom g = make_geom(hidden_k, dim_n);
    const int ku = (divUp(g.pw4, 2) + 3) & ~3, sh = divUp(ku, 64);
    if (hidden_k % 32 || sh != 3 || ku - 128 <= 32 || ku * 2 < g.pw4 || divUp(g.pw4, 64) > FP_ROWS) return false;
    const int nc2 = divUp(dim_n, nb);
    if (dim_n / nb < 1 || nc2 > FP_NC2MAX || (size_t)divUp(nc2 * g.pw4 * 16, 1024) * 1024 > FfnPairLds::DW_BYTES) return false;
    return nc2 * g.sh * 2 <= 3072 && nc2 * g.pzh * 4 <= 1024;
}

}  // namespace q4
