// gemv_q4.h -- the INT4-AWQ dequant-fused GEMV for gfx950 (wave64). All three launch shapes of the
// reference share this one kernel template:
//   MODE_PLAIN : mat_vec_kernel_int4      gpu_kernels.h:213-240  (accum / KV addressing)
//   MODE_QKV   : qkv_matvec_kernel        gpu_kernels.h:242-254  (+ optional RoPE epilogue :332-355)
//   MODE_FFN   : ffn_matvec_silu_kernel   gpu_kernels.h:256-275
// and can fold rmsnorm (gpu_kernels.h:72-105) into the activation staging.
//
// Mapping (designed from the data layout, SURVEY a1, not from the reference's 32x4 blocks):
//  * a column n is a contiguous run of K/32 uint4 (32 nibbles each). One WAVE owns COLS columns; lane l reads
//    uint4 j = slot*64 + l of each with one 128-bit non-temporal buffer load: a wave instruction moves 1 KiB of
//    one column, fully coalesced; the column offset rides in the SGPR soffset, so loads cost no VALU. Every load
//    of a wave is issued before anything is consumed: the whole matrix is in flight at once.
//  * the activation vector is staged once per block into LDS, pre-permuted so that the 16 B a lane needs for
//    weight dword d are one conflict-free ds_read_b128: unit [(slot*4 + d)*64 + lane] holds (x0,x4),(x1,x5),
//    (x2,x6),(x3,x7) of the 8 inputs that dword multiplies.
//  * dequant without converts, subtracts or magic exponents -- the kernel is VALU-bound if written naively (measured:
//    v_and_or, v_dot2c, v_pk_*, v_cvt_* issue at half rate, ~1.8 ns per wave instruction per SIMD; plain VOP2 integer
//    ops at full rate): a nibble left in place IS an fp16 denormal, (w & 0x000F000F) = the half2 (q_i, q_{i+4}) * 2^-24,
//    and v_dot2c_f32_f16 multiplies denormal inputs exactly (verified on gfx950, tools/lab/t_denorm.hip). The odd nibbles
//    sit 4 bits higher (16 q * 2^-24). Two fp32 accumulators, acc_e and acc_o, then
//        sum_k q x = 2^20 (16 acc_e + acc_o),      sum_k (q - z) x = that - z * (sum of the 32 x)
//    with the x-only sum computed once per block at staging and the 2^20 applied once per column at the end (all
//    power-of-two scalings are exact). 1 shift + 4 v_and_b32 (full rate) + 4 v_dot2c per 8 weights; products are
//    exact in fp32 (4 x 11 bits), so the arithmetic is as accurate as the reference's (q - z) * s * x in fp32.
//  * cross-lane reduction transposes while it reduces (v_permlane32_swap, v_permlane16_swap, DPP row ops):
//    4 (8) column sums cost 10 (20) VALU instead of 44 (88), and leave column r's total in DPP row r, so the
//    epilogue (residual add / SiLU / RoPE) runs once for all columns in parallel lanes.
// No MFMA: 3.84 flop/B, the kernel is an HBM stream.
#pragma once
#include <type_traits>
#include "q4_device.h"
#include "q4_internal.h"

#ifndef Q4_ZS_AUX
#define Q4_ZS_AUX 0  // zeros / scales: lines are shared by neighbouring lanes and waves, default policy
#endif
#ifndef Q4_W_AUX
#define Q4_W_AUX 2   // buffer-load cache policy of the weight stream: nt (read once per token)
#endif

namespace q4 {

constexpr int MODE_PLAIN = 0, MODE_QKV = 1, MODE_FFN = 2;

struct GemvMat {
    const uint32_t* w;
    const uint32_t* z;
    const q4_half* s;
};

struct GemvArgs {
    GemvMat m[3];
    q4_half* out[3];
    const q4_half* x;
    const q4_half* rms_w;   // non-null: x is the raw residual, rmsnorm it while staging
    const int* pPos;
    int K, N;               // input length, output columns (per matrix)
    int N_kv;               // QKV with grouped-query attention: columns of the k and v matrices (0: same as N)
    int pw4, pzh, sh, nslots;
    int ku;                 // K split (KS > 1): uint4 units per k-part, a multiple of 4 (whole groups); part p owns units [p*ku, (p+1)*ku)
    int accum;              // PLAIN: out = half(float(out) + sum)
    long long loff;         // PLAIN: -1 = none. QKV: KV-cache layer offset (64-bit: 13B at 16K positions exceeds 2^31 halves)
    int rope;               // QKV: rotate q and k in the epilogue
    int head_size;
    float rope_theta;
    unsigned long long* dbg;   // ABL == 3 only: per-wave s_memtime stamps
    const float2* rope_table;  // [seq_len][head_size/2] (cos, sin) built with the reference's formula; null: compute
    int early;                 // waves in the first `early` slots of a SIMD issue their weight loads before the staging ends
    unsigned* bump;            // QKV: epoch word of the FOLLOWING attention + o-proj launch, advanced once by block (0, 0)
};

// In-launch hand-off (layer_attn.hip: attention -> o-proj as ONE launch) with data-tagged granules: a vector
// crosses CUs as 8-byte {two halves, tag} words, each written by ONE write-through (sc1) store, so a word is either old or
// complete and carries its own validity -- no flag, no drain, no arrival counter on the producer side (measured: sc1
// payload -> s_waitcnt vmcnt(0) -> returning atomic cost every producer block 1.7 us at its end). The tag is the launch's
// epoch (a device word read at entry; it is bumped by the PRECEDING launch of the stream -- the fused QKV GEMV -- so every
// block of the launch reads the same value whatever the dispatch order or timing), so stale granules of earlier launches
// never match. Consumers put their weight loads in flight first, then poll with uncached (sc1) loads, bounded, with s_sleep.
// Polls must stay off shared lines: an uncached poll lands on the memory channel of its address, and since a wave's loads
// return in order, 160 blocks polling two shared cache lines stalled every weight stream of the launch (the later QKV
// blocks ended 4-10 us late, tools/lab/timeline_block.py). Here a line is polled by at most a handful of blocks.
// (MI355X_MICROARCH.md: "handoff-1to1 ... data-tagged granules", R2.) ROLE_NONE compiles to the stand-alone kernel.
constexpr int ROLE_NONE = 0, ROLE_CONSUMER = 2;
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
struct Handoff {
    unsigned tag;          // this launch's tag = the epoch word as the preceding launch left it (>= 1: zeroed buffers never match)
    u32x2v* pub;           // producer (attention role): granule vector to publish into ([dim/2])
    const u32x2v* sub;     // consumer: granule vector to read (o-proj: the attention output, [K/2])
    int sentinel;          // consumer: granule polled first (staggered over the blocks: few pollers per line)
    unsigned* error;       // set to 1 when a bounded poll ran out (results are garbage then; the host reports it)
    unsigned dead;         // the error word as read at entry, kept raw (arithmetic on it at entry would pull the load's wait there).
                           // != 0: an earlier launch timed out and the host has not cleared the state yet -- results are void anyway,
                           // so nobody spins again: a long queue of launches behind a failure drains at once
    unsigned long long hold_until;   // consumer: wall clock (100 MHz) before which the role does not request its weights, or 0 (layer_attn.h: split-context forms)
    bool mute;             // profiling build: this attention block does not publish (provokes a real time-out, tests/prof_cases.py)
    unsigned long long* stamp;   // profiling build: wall clock right after the wait
};
constexpr unsigned POLL_LIMIT = 1u << 20;   // ~0.25 us per poll: give up after ~0.3 s instead of hanging the GPU

// Where granule g of the residual stream's hand-off vectors lies (gemv_ffn_pair.h: x between the FFN half and the next QKV, level 6's x): records of GRAN_REC
// granules (64 bytes -- what one block publishes with one store instruction), one record every GRAN_STRIDE granules (2 KiB). Dense, a 4096-wide vector is 128
// lines of 16 KB that all 256 CUs read at once -- two blocks on different XCDs write into each line, a handful of memory channels serve 32 K requests a pass.
// Spread over 512 KB: record every 64 / 128 / 256 / 512 / 1024 / 2048 / 4096 B: 1021.0 / 1019.2 / 1027.8 / 1024.6 / 1025.9 / 1030.3 / 1026.1 tokens/s.
// (The attention output in such half-line records measured -0.3 % / -0.5 %: its units publish whole lines -- line_slot below. EXPERIMENTS.md #48)
constexpr unsigned GRAN_REC = 8u, GRAN_STRIDE = 256u;
__host__ __device__ constexpr unsigned gran_slot(unsigned g) { return (g / GRAN_REC) * GRAN_STRIDE + (g % GRAN_REC); }
__host__ __device__ constexpr size_t gran_area_words(size_t granules) { return (granules + GRAN_REC - 1) / GRAN_REC * GRAN_STRIDE * 2; }   // 32-bit words of a spread vector
// The attention output: whole 128-byte lines (16 granules: what a (head, V slice) unit publishes with one store) one per 2 KiB: +0.2 % (-n 256), +0.1 % (-n 2048).
// (The hb vector as lines one per 2 KiB measured level: it stays dense -- its first gather pass goes through the XCDs' L2s.)
__host__ __device__ constexpr unsigned line_slot(unsigned g) { return (g >> 4) * GRAN_STRIDE + (g & 15u); }
__host__ __device__ constexpr size_t line_area_words(size_t granules) { return (granules + 15) / 16 * GRAN_STRIDE * 2; }
__device__ __forceinline__ void store_granule(u32x2v* p, unsigned data, unsigned tag) {
    const u32x2v g = {data, tag};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ u32x2v load_granule(const u32x2v* base, unsigned index) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b64(r, index * 8u, 0, 16);   // sc1: served past the L1
}
__device__ __forceinline__ u32x4 load_granule2(const u32x2v* base, unsigned index) {   // two consecutive granules
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, index * 8u, 0, 16);
}
template <int MODE>
struct ModeTraits { static constexpr int NMAT = (MODE == MODE_FFN) ? 2 : 1; };

// permute 8 consecutive halves (x0..x7 as 4 dwords) into (x0,x4),(x1,x5),(x2,x6),(x3,x7)
__device__ __forceinline__ u32x4 permute_x8(u32x4 v) {
    u32x4 o;
    o[0] = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u);   // lo16(v0) | lo16(v2)<<16
    o[1] = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);   // hi16(v0) | hi16(v2)<<16
    o[2] = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u);
    o[3] = __builtin_amdgcn_perm(v[3], v[1], 0x07060302u);
    return o;
}

// vdst/src half exchange (v_permlane32_swap): returns a' + b' where lanes 0-31 = a[l] + a[l+32],
// lanes 32-63 = b[l-32] + b[l]. Inline asm on purpose: with __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) folded
// the two results of the builtin into one register here (emitted v_add v, v14, v14 after `v_permlane32_swap v14, v7`),
// i.e. 2*a' instead of a' + b'. The s_nop covers the VALU-write -> permlane-read hazard (2 wait states).
__device__ __forceinline__ float swap32_add(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// same at 16-lane granularity (v_permlane16_swap): even rows = a[l] + a[l+16], odd rows = b[l-16] + b[l]
__device__ __forceinline__ float swap16_add(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// 4 per-lane partial sums -> every lane of DPP row r (lanes 16r..16r+15) holds the 64-lane total of v[r]
__device__ __forceinline__ float reduce4_rows(float v0, float v1, float v2, float v3) {
    const float u0 = swap32_add(v0, v2);   // lower half: v0, upper half: v2
    const float u1 = swap32_add(v1, v3);   // lower half: v1, upper half: v3
    return row16_sum(swap16_add(u0, u1));  // rows: v0, v1, v2, v3
}

// The accumulate that does not round to nearest. v_dot2c_f32_f16 (D = a.x * b.x + a.y * b.y + D) aligns its three addends and TRUNCATES toward minus
// infinity: on this kernel family's operands (denormal nibbles x fp16 inputs + a running sum) tools/lab/t_dot2_round.hip measures a mean error of -0.3 ..
// -0.4 fp32 ulp per instruction where round-to-nearest has 0. 2048 such instructions per output add up to a signed error of about -4e-6 per GEMV output
// (-2e-5 for the down projection's inputs, tools/bias_probe.py), the same for every output: a coherent offset that the residual stream accumulates
// (-4.6e-5 per layer) and that weights with a non-zero column mean turn into 7 % more distance from the exact forward than the reference's own
// arithmetic has after 32 layers (tools/error_growth.py; round 5 found it, VERDICT r04 asked). The cure costs nothing in the inner loop: the staging
// stores ODD uint4 units of x negated (and their x-only sum with them); a lane's units all have its own parity (unit = 64 slot + lane, K-split bases are
// multiples of 4), so an odd lane accumulates the NEGATED column sum, on which the same truncation pushes the other way, and gets its sign back in
// the wave reduction. Even and odd lanes then err in opposite directions and the offset cancels.
// In code: a thread stages chunks of units with the parity of (tid >> 2) (blocks are multiples of 8 threads wide), so it negates what it stages --
// through the sign of the rmsnorm scale where the kernel normalises (free), else by flipping the halves' sign bits -- and the x-only sum, taken
// from the staged values, follows by itself. The wave reduction gives the signs back for free as well: the DPP steps that combine lanes of
// opposite parity (lane ^ 1, the two mirrors) subtract instead of add, so even lanes end with the total (the writers are lanes 0, 16, 32, 48).
__device__ __forceinline__ unsigned q4_stage_sign_bits(unsigned tid) { return (tid & 4u) ? 0x80008000u : 0u; }
__device__ __forceinline__ float q4_signed_scale(float ss, unsigned sign_bits) { return as_f(as_i(ss) ^ (int)(sign_bits & 0x80000000u)); }
__device__ __forceinline__ u32x4 q4_signed_x(u32x4 v, unsigned sign_bits) { return (u32x4){v[0] ^ sign_bits, v[1] ^ sign_bits, v[2] ^ sign_bits, v[3] ^ sign_bits}; }
// four per-lane column sums of an int4 GEMV (odd lanes hold them negated) -> every EVEN lane of DPP row r holds the 64-lane total of v[r]
__device__ __forceinline__ float reduce4_q4(float v0, float v1, float v2, float v3) {
    const float u0 = swap32_add(v0, v2);   // (lanes l and l + 32, l and l + 16: the same parity)
    const float u1 = swap32_add(v1, v3);
    float v = swap16_add(u0, u1);
    v -= dpp_mov<0xB1>(v);    // lane ^ 1: the other parity
    v += dpp_mov<0x4E>(v);    // lane ^ 2
    v -= dpp_mov<0x141>(v);   // row_half_mirror: i <-> 7 - i, the other parity
    v -= dpp_mov<0x140>(v);   // row_mirror: i <-> 15 - i, the other parity
    return v;
}

// ABL (profiling-only ablation builds): 0 = product, 1 = loads kept but no dequant math, 2 = math on constants
// without the weight loads.
// register-heavy shapes (>= 24 uint4 in flight per lane, e.g. the 7B down projection) run 4 waves per block with
// the whole 512-register file per lane; the others may use 8 waves per block
// order in which the dispatcher filled this CU: HW_ID.wave_id (slot on the SIMD, [3:0]) major, simd_id ([5:4]) minor
__device__ __forceinline__ int early_rank() {
    const unsigned id = __builtin_amdgcn_s_getreg(10244);   // hwreg(HW_REG_HW_ID, 0, 6)
    return (int)(((id & 15u) << 2) | (id >> 4));
}

template <int MODE, int SLOTS, int COLS, int KS = 1>
struct LaunchTraits { static constexpr int MAX_THREADS = (ModeTraits<MODE>::NMAT * SLOTS * COLS >= 24) ? 256 : 512; };

// LDS of one block: permuted x [rows][4][64] x 16 B, the x-only sums [rows][64], and `part` (rmsnorm chunk partials: one float
// per 16-byte unit; K-split kernels only exchange a few totals there). K-split kernels carry one extra all-zero row: the idle
// lanes of a part's last slot read it (their neighbours in LDS are the NEXT part's inputs, not padding).
template <int SLOTS, int KS>
struct LdsLayout {
    static constexpr int TS = SLOTS * KS;
    static constexpr int ROWS = KS > 1 ? TS + 1 : TS;
    static constexpr int NUNITS = ROWS * 256;
    static constexpr size_t XS = (size_t)NUNITS * 16, SX = (size_t)ROWS * 256, PART = KS > 1 ? 1024 : (size_t)NUNITS * 4;
    static constexpr size_t BYTES = XS + SX + PART + 16;
};

// KS = 2 (PLAIN only): two waves share a column group, each takes SLOTS of the KS*SLOTS k-slots and the partial
// totals meet in LDS. The N = dim projections (o, down) have only N/4 = 1024 column groups = one wave per SIMD, and a
// single wave issues one VALU instruction per ~5.5 cycles (measured) -- the ~1000-instruction down-projection wave
// was a 3.4 us serial chain; splitting K doubles the waves and halves that chain.
// HALF: the column's last k-slot holds at most 32 uint4 (K = 5120: 160 = 2 x 64 + 32). Instead of running that slot with
// half of the lanes multiplying zero padding, lanes 0-31 take it for column c and lanes 32-63 for column c + 1 of each
// column pair: one load and one dequant-dot evaluation per PAIR (13B q/k/v/o/gate/up: 12 -> 10 per wave).
template <int MODE, int SLOTS, int COLS, bool NORM, int ABL, int KS, bool HALF, int ROLE>
__device__ __forceinline__ void gemv_q4_body(const GemvArgs& a, const unsigned vbx, const unsigned vby, const Handoff& ho) {
    static_assert(ROLE == ROLE_NONE || (KS == 1 && ABL != 3), "hand-off roles: no K split, no time stamps");
    static_assert(!HALF || COLS % 2 == 0, "shared half slot: column pairs");
    constexpr int NMAT = ModeTraits<MODE>::NMAT;
    constexpr int NV = NMAT * COLS;             // column sums per wave
    static_assert(NV == 4 || NV == 8 || (NV == 2 && MODE == MODE_PLAIN), "row-distributed epilogue handles 4 or 8 sums per wave (plain: 2 as well)");
    static_assert(KS == 1 || (KS == 2 && MODE == MODE_PLAIN && COLS == 4 && !NORM), "K split: plain GEMV, 4 columns");
    using Lds = LdsLayout<SLOTS, KS>;
    constexpr int TS = Lds::ROWS;               // 64-unit rows staged in LDS: the column's k-slots (+ the zero row of a K split)
    constexpr int NUNITS = Lds::NUNITS;         // 16-byte LDS units (zero padded past K)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                                   // [TS][4][64] permuted x
    float* sx = reinterpret_cast<float*>(smem + Lds::XS);                         // [TS][64] -(sum of the 32 x) * 2^-20
    float* part = reinterpret_cast<float*>(smem + Lds::XS + Lds::SX);             // chunk partials / K-split exchange

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: column offsets stay in SGPRs
    const int nw = blockDim.x >> 6;
    const int wg = vbx * (nw / KS) + wave / KS;   // global column-group index
    const int khalf = wave % KS;                // which k-part of the column this wave sums
    // K split: part p owns uint4 units [p * ku, (p + 1) * ku) -- balanced to whole quantisation groups, not to 64-unit slots
    const unsigned ubase = KS > 1 ? (unsigned)(khalf * a.ku) : 0u;
    const unsigned uend = KS > 1 ? (ubase + (unsigned)a.ku < (unsigned)a.pw4 ? ubase + (unsigned)a.ku : (unsigned)a.pw4) : (unsigned)a.pw4;
    const unsigned nchunks = (unsigned)a.K >> 3;            // real 8-half chunks
    const int mat0 = (MODE == MODE_QKV) ? vby : 0;
    // grouped-query attention: k and v have kv_dim < dim columns; the launch grid is sized for q, surplus blocks leave
    const int N = (MODE == MODE_QKV && mat0 != 0 && a.N_kv > 0) ? a.N_kv : a.N;
    if (MODE == MODE_QKV && (int)(vbx * (blockDim.x >> 6)) * COLS >= N) return;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ABL == 3) { ts[0] = __builtin_readcyclecounter(); ts[6] = wall_clock64(); }   // [6]: 100 MHz, same on every XCD

    // ---- column ownership ---------------------------------------------------------------------
    int col[COLS];
    if (MODE == MODE_QKV) {
        // pairs (i, i + head_size/2) of one head live in the same wave so RoPE needs no other wave
        constexpr int P = COLS / 2;
        const int hp = a.head_size >> 1;
#pragma unroll
        for (int pi = 0; pi < P; pi++) {
            const int p = wg * P + pi;
            const int head = p / hp, i = p - head * hp;
            col[pi] = head * a.head_size + i;
            col[P + pi] = col[pi] + hp;
        }
    } else {
#pragma unroll
        for (int c = 0; c < COLS; c++) col[c] = wg * COLS + c;
    }
    int colc[COLS];   // clamped: loads stay in bounds, the store is skipped
#pragma unroll
    for (int c = 0; c < COLS; c++) colc[c] = col[c] < N ? col[c] : N - 1;

    // the position is fetched in front of everything else: hoisted next to its use, hipcc puts a vector load +
    // vmcnt(0) between the weight loads and the first dot product, and the per-slot waits are gone
    int pos_now = 0;
    if (MODE == MODE_QKV) { if (mat0 != 0 || a.rope) pos_now = *a.pPos; }
    else if (MODE == MODE_PLAIN) { if (a.loff != -1) pos_now = *a.pPos; }
    asm volatile("" : "+v"(pos_now));

    // ---- 1. activation loads first: their wait (counted vmcnt) leaves the weight loads in flight ----
    u32x4 xraw[TS], wraw[TS];
    auto load_x = [&]() {
#pragma unroll
        for (int i = 0; i < TS; i++) {
            const unsigned u = tid + i * blockDim.x;
            const unsigned uc = u < nchunks ? u : nchunks - 1;  // clamped, branch-free
            xraw[i] = reinterpret_cast<const u32x4*>(a.x)[uc];
            if (NORM) wraw[i] = reinterpret_cast<const u32x4*>(a.rms_w)[uc];
        }
    };
    // consumer role: x arrives as granules ({2 halves, tag} per 8 bytes, 4 granules per 8-half chunk) written inside this
    // launch by other CUs. Every thread re-reads ITS chunks until they carry the launch's tag (a chunk that is complete is
    // not read again); false if the bounded retry ran out.
    auto load_x_granules = [&]() -> bool {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < TS; i++) {
            const unsigned u = tid + i * blockDim.x;
            const unsigned uc = u < nchunks ? u : nchunks - 1;
            for (unsigned tries = 0;; tries++) {
                const u32x4 g01 = load_granule2(ho.sub, line_slot(uc * 4)), g23 = load_granule2(ho.sub, line_slot(uc * 4 + 2));
                xraw[i] = (u32x4){g01[0], g01[2], g23[0], g23[2]};
                if (g01[1] == ho.tag && g01[3] == ho.tag && g23[1] == ho.tag && g23[3] == ho.tag) break;
                if (tries >= POLL_LIMIT / 4 || ho.dead != 0u) { ok = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        return ok;
    };
    if (ROLE == ROLE_CONSUMER && NORM) {   // the norm weights do not depend on the producers
#pragma unroll
        for (int i = 0; i < TS; i++) {
            const unsigned u = tid + i * blockDim.x;
            wraw[i] = reinterpret_cast<const u32x4*>(a.rms_w)[u < nchunks ? u : nchunks - 1];
        }
    }
    if (ROLE != ROLE_CONSUMER) load_x();

    // ---- 2. weight / zero / scale loads: the first PRE slots go out BEFORE the activation staging (so HBM is
    // busy while the block normalises x), the rest right after it. Measured with s_memtime stamps: issuing every
    // load first parks a wave ~7000 cycles in VMEM issue back-pressure and only then starts the ~4000-cycle
    // x -> norm -> LDS chain; staging in the shadow of the first batch removes that chain from the kernel's tail.
    // Buffer loads: descriptor per tensor (SGPRs), wave-uniform column byte offset in soffset (SGPR), the lane's
    // offset inside the column in voffset (one VGPR per slot, shared by every column and matrix) -> no per-load
    // address arithmetic on the VALU. aux = 2 is the non-temporal hint (weights are read once per token).
    // measured on MI355X (tools/lab/sweep_gemv.py): for K <= 8192 staging FIRST wins (QKV 8.1 -> 7.4 us, gate 6.3 -> 5.7 us):
    // the x chain is short when no weight request is queued ahead of it; for the long-K down projection
    // (6-7 slots, 1 wave per SIMD) half of the loads in front of the staging hides it better (7.5 -> 7.2 us)
    // ABL == 5 is not an ablation but the "all loads first" order: when the grid is about one block per CU (o-proj:
    // 256 blocks) nothing else queues on the CU, so x chain and weight latency overlap (4.3 -> 4.0 us in-graph)
    constexpr int PRE = ABL == 5 ? SLOTS : (SLOTS >= 6 ? (SLOTS + 1) / 2 : 0);
    u32x4 W[NMAT][SLOTS][COLS];
    unsigned ZW[NMAT][SLOTS][COLS];
    uint16_t SC[NMAT][SLOTS][COLS];
    __amdgpu_buffer_rsrc_t rw[NMAT], rz[NMAT], rs[NMAT];
#pragma unroll
    for (int m = 0; m < NMAT; m++) {
        const GemvMat& M = a.m[mat0 + m];
        rw[m] = __builtin_amdgcn_make_buffer_rsrc((void*)M.w, 0, N * a.pw4 * 16, 0x00020000);
        rz[m] = __builtin_amdgcn_make_buffer_rsrc((void*)M.z, 0, N * a.pzh * 4, 0x00020000);
        rs[m] = __builtin_amdgcn_make_buffer_rsrc((void*)M.s, 0, N * a.sh * 2, 0x00020000);
    }
#define Q4_ISSUE_SLOT(s)                                                                                          \
    {                                                                                                             \
        const unsigned j = ubase + (s) * 64 + lane;                                                               \
        const unsigned jj = j < uend ? j : uend - 1; /* tail lanes re-read the part's last unit */                \
        _Pragma("unroll") for (int m = 0; m < NMAT; m++) _Pragma("unroll") for (int c = 0; c < COLS; c++) {       \
            ZW[m][s][c] = __builtin_amdgcn_raw_buffer_load_b32(rz[m], (jj >> 5) * 4, colc[c] * a.pzh * 4, Q4_ZS_AUX); \
            SC[m][s][c] = __builtin_amdgcn_raw_buffer_load_b16(rs[m], (jj >> 2) * 2, colc[c] * a.sh * 2, Q4_ZS_AUX);  \
            if (ABL == 2)                                                                                         \
                W[m][s][c] = (u32x4){jj * 2654435761u, jj ^ 0x9E3779B9u, (unsigned)colc[c] * 40503u, jj + 7u};    \
            else                                                                                                  \
                W[m][s][c] = __builtin_amdgcn_raw_buffer_load_b128(rw[m], jj * 16, colc[c] * a.pw4 * 16, Q4_W_AUX); \
        }                                                                                                         \
    }
    // the shared half slot: column offset per lane (lanes 32-63 belong to the pair's second column), results kept in
    // the pair's first entries W/ZW/SC[m][SLOTS-1][2p]
    const bool upper = lane >= 32u;
#define Q4_ISSUE_HALF()                                                                                           \
    {                                                                                                             \
        const unsigned jh = ubase + (SLOTS - 1) * 64 + (lane & 31u);                                              \
        const unsigned jj = jh < uend ? jh : uend - 1;                                                            \
        _Pragma("unroll") for (int m = 0; m < NMAT; m++) _Pragma("unroll") for (int c = 0; c < COLS; c += 2) {    \
            const unsigned cw = upper ? (unsigned)colc[c + 1] : (unsigned)colc[c];                                \
            ZW[m][SLOTS - 1][c] = __builtin_amdgcn_raw_buffer_load_b32(rz[m], (jj >> 5) * 4 + cw * a.pzh * 4, 0, 0); \
            SC[m][SLOTS - 1][c] = __builtin_amdgcn_raw_buffer_load_b16(rs[m], (jj >> 2) * 2 + cw * a.sh * 2, 0, 0);  \
            W[m][SLOTS - 1][c] = __builtin_amdgcn_raw_buffer_load_b128(rw[m], jj * 16 + cw * a.pw4 * 16, 0, 2);   \
        }                                                                                                         \
    }
#define Q4_ISSUE_ANY(s) { if (HALF && (s) == SLOTS - 1) Q4_ISSUE_HALF() else Q4_ISSUE_SLOT(s) }
    if (ROLE == ROLE_CONSUMER) {   // held back while the producers' stream needs the memory system for itself (layer_attn.h)
        if (ho.hold_until != 0ull && ho.dead == 0u) while (wall_clock64() < ho.hold_until) __builtin_amdgcn_s_sleep(4);   // (a launch queued behind a time-out holds nothing back)
    }
#pragma unroll
    for (int s = 0; s < PRE; s++) Q4_ISSUE_ANY(s)
    __builtin_amdgcn_sched_barrier(0);

    if (ABL == 3) ts[1] = __builtin_readcyclecounter();
    // ---- 3. stage the activation vector into LDS, optional fused rmsnorm -----------------------
    // Early birds: every wave of a launch enters within ~0.5 us and the CU's vector-memory path returns data in issue
    // order, so normally the weight loads wait until x is staged (they would otherwise sit in front of other waves'
    // x loads, or park this wave in issue back-pressure). The first `a.early` waves the dispatcher placed on the CU (HW_ID rank): once their own x has arrived (by then every other wave's x loads
    // are queued too) they put their weight loads in flight and only then finish the staging, so HBM streams during
    // the ~1.9 us x chain of the launch. Kept to ~64 KB per CU: those loads are accepted without back-pressure.
    const bool early = ABL != 4 && ABL != 5 && PRE < SLOTS && a.early > 0 && early_rank() < (a.early & 255);
    // (single load site in program order, the staging tail before OR after it: after a two-path join with loads on
    // both sides hipcc's waitcnt pass falls back to vmcnt(0) and the progressive per-slot waits are lost)
    auto stage_tail = [&]() {
        float ss = 1.f;
        if (NORM) {
#pragma unroll
            for (int i = 0; i < TS; i++) {
                const unsigned u = tid + i * blockDim.x;
                if (u < NUNITS) part[u] = u < nchunks ? sumsq8(xraw[i], 0.f) : 0.f;
            }
            __syncthreads();
            if (ABL == 3) ts[2] = __builtin_readcyclecounter();
            ss = rms_scale_from_partials<NUNITS>(part, NUNITS, a.K);
        }
        const h2 ones = {(f16_t)1.0f, (f16_t)1.0f};
        const unsigned sgn = q4_stage_sign_bits(tid);      // odd units are staged negated (q4_stage_sign_bits)
        if (NORM) ss = q4_signed_scale(ss, sgn);
#pragma unroll
        for (int i = 0; i < TS; i++) {
            const unsigned u = tid + i * blockDim.x;
            u32x4 v = xraw[i];
            if (NORM) v = rms_apply8(v, wraw[i], ss);     // (ss carries this thread's unit sign)
            else v = q4_signed_x(v, sgn);
            if (u >= nchunks) v = (u32x4){0u, 0u, 0u, 0u};
            const u32x4 pv = permute_x8(v);
            // x-only term of the zero point: sum of the 32 inputs of uint4 j (4 consecutive units = one lane quad)
            float cb = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; d4++) cb = __builtin_amdgcn_fdot2(as_h2(pv[d4]), ones, cb, false);
            cb += dpp_mov<0xB1>(cb); cb += dpp_mov<0x4E>(cb);   // quad sum
            const unsigned j = u >> 2, d = u & 3u;
            if (u < NUNITS) {
                xs[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = pv;
                if (d == 0) sx[j] = cb * -9.5367431640625e-07f;   // -(sum x) * 2^-20
            }
        }
        __syncthreads();
        if (ABL == 3) ts[3] = __builtin_readcyclecounter();
    };
    if (early) {    // give the waves that entered last time to queue their x loads (bits 8+ of a.early, 128-cycle steps)
        for (int i = a.early >> 8; i > 0; i--) __builtin_amdgcn_s_sleep(2);
    }
    if (ROLE == ROLE_CONSUMER) {   // every weight load is in flight (PRE == SLOTS): now wait for the producers' granules
        static_assert(ROLE != ROLE_CONSUMER || ABL == 5, "consumer role: all loads first");
        if (tid == 0 && ho.dead == 0u) {   // one lane polls ONE granule (spread over lines: few pollers per line) ...
            unsigned i = 0;
            while (load_granule(ho.sub, line_slot((unsigned)ho.sentinel))[1] != ho.tag && ++i < POLL_LIMIT) __builtin_amdgcn_s_sleep(2);
            if (i >= POLL_LIMIT) __hip_atomic_store(ho.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // ... then the block reads the whole vector; every granule validates itself
        if (!load_x_granules() && ho.dead == 0u) __hip_atomic_store(ho.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef Q4_PROFILING
        if (ho.stamp && tid == 0) *ho.stamp = wall_clock64();
#endif
    }
    if (ABL != 4 && !early) stage_tail();      // (ABL 4: no staging at all, garbage x -- the kernel without the x chain)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = PRE; s < SLOTS; s++) Q4_ISSUE_ANY(s)
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 4 && early) stage_tail();
#undef Q4_ISSUE_ANY
#undef Q4_ISSUE_HALF
#undef Q4_ISSUE_SLOT


    // ---- 4. consume in issue order -------------------------------------------------------------
    float colsum[NMAT][COLS];
#pragma unroll
    for (int m = 0; m < NMAT; m++)
#pragma unroll
        for (int c = 0; c < COLS; c++) colsum[m][c] = 0.f;

#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        const bool hs = HALF && s == SLOTS - 1;                 // shared half slot: both halves of the wave read units 0-31
        const unsigned lu = hs ? (lane & 31u) : lane;
        unsigned j = ubase + s * 64 + lu;                       // the uint4 unit this lane multiplies in this slot
        unsigned jx = j;                                        // ... and where its inputs sit in LDS
        if (KS > 1) {                                           // past the part's end: the all-zero row (the unit was re-read, see above)
            jx = j < uend ? j : (unsigned)((TS - 1) * 64);
            j = j < uend ? j : uend - 1;
        }
        u32x4 X[4];
        const unsigned xrow = KS > 1 ? ((jx >> 6) << 8) + (jx & 63u) : (unsigned)((s * 4) << 6) + lu;
#pragma unroll
        for (int d = 0; d < 4; d++) X[d] = xs[xrow + (d << 6)];
        const float corr = sx[KS > 1 ? jx : (unsigned)(s * 64) + lu];
        const unsigned zsh = ((j >> 2) & 7u) * 4u;
#pragma unroll
        for (int m = 0; m < NMAT; m++)
#pragma unroll
            for (int c = 0; c < COLS; c += (hs ? 2 : 1)) {
                float t;
                if (ABL == 1) {
                    const u32x4 w = W[m][s][c];
                    t = as_f(((w[0] ^ w[1] ^ w[2] ^ w[3]) & 0x007FFFFFu) | 0x3F000000u) + as_f(X[s & 3][c & 3] & 0x3FFFFFFFu);
                } else {
                    const u32x4 w = W[m][s][c];
                    float acc_e = 0.f, acc_o = 0.f;
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const unsigned ww = w[d];
                        const unsigned tt = ww >> 8;
                        acc_e = __builtin_amdgcn_fdot2(as_h2(ww & 0x000F000Fu), as_h2(X[d][0]), acc_e, false);   // (n0,n4) * 2^-24
                        acc_o = __builtin_amdgcn_fdot2(as_h2(ww & 0x00F000F0u), as_h2(X[d][1]), acc_o, false);   // (n1,n5) * 2^-20
                        acc_e = __builtin_amdgcn_fdot2(as_h2(tt & 0x000F000Fu), as_h2(X[d][2]), acc_e, false);   // (n2,n6)
                        acc_o = __builtin_amdgcn_fdot2(as_h2(tt & 0x00F000F0u), as_h2(X[d][3]), acc_o, false);   // (n3,n7)
                    }
                    const float zf = (float)((ZW[m][s][c] >> zsh) & 0xFu);
                    t = __builtin_fmaf(acc_e, 16.f, acc_o);          // 2^-20 * sum q x
                    t = __builtin_fmaf(zf, corr, t);                 // - z * 2^-20 * sum x
                }
                // idle tail lanes (j >= pw4) multiply re-read weights by the zero padding of xs/sx: exactly 0
                if (hs) {                                       // lanes 0-31 worked for column c, lanes 32-63 for column c + 1
                    const float v = h2f(SC[m][s][c]) * t;
                    colsum[m][c] += upper ? 0.f : v;
                    colsum[m][c + 1] += upper ? v : 0.f;
                } else {
                    colsum[m][c] = __builtin_fmaf(h2f(SC[m][s][c]), t, colsum[m][c]);
                }
            }
        if (ABL == 3 && s < 3) { asm volatile("" : "+v"(colsum[0][0])); ts[4 + s] = __builtin_readcyclecounter(); }
    }

    // ---- 5. transposing wave reduction + lane-parallel epilogue ---------------------------------
    const int row = lane >> 4;                 // DPP row of this lane = which column's total it will hold
    const bool writer = (lane & 15u) == 0;

    if constexpr (MODE == MODE_PLAIN) {
        q4_half* out = a.out[0];
        if (a.loff != -1) out += (size_t)a.loff + (size_t)pos_now * N;             // gpu_kernels.h:225-227
        if constexpr (COLS == 4) {
            float tot = reduce4_q4(colsum[0][0], colsum[0][1], colsum[0][2], colsum[0][3]) * 1048576.f;   // 2^20, exact
            const int n = wg * 4 + row;
            if (KS > 1) {                       // fixed order: k-parts from the lowest up
                if (writer) part[wave * 4 + row] = tot;
                __syncthreads();
                if (khalf == 0) tot += part[(wave + 1) * 4 + row];
            }
            if (writer && khalf == 0 && n < N) {
                float r = tot;
                if (a.accum) r += h2f(out[n]);                                      // :229-230
                out[n] = f2h(r);                                                    // :231
            }
        } else if constexpr (COLS == 2) {   // rows 0 and 1 hold the two columns (the same pair sums as in the four-column form)
            const float tot = reduce4_q4(colsum[0][0], colsum[0][1], 0.f, 0.f) * 1048576.f;
            const int n = wg * 2 + row;
            if (writer && row < 2 && n < N) {
                float r = tot;
                if (a.accum) r += h2f(out[n]);
                out[n] = f2h(r);
            }
        } else {   // COLS == 8: row r holds columns 2r (w0) and 2r+1 (w1)
            const float w0 = reduce4_q4(colsum[0][0], colsum[0][2], colsum[0][4], colsum[0][6]) * 1048576.f;
            const float w1 = reduce4_q4(colsum[0][1], colsum[0][3], colsum[0][5], colsum[0][7]) * 1048576.f;
            const int n = wg * 8 + row * 2;
            if (writer && n < N) {
                float r0 = w0, r1 = w1;
                if (a.accum) { r0 += h2f(out[n]); r1 += h2f(out[n + 1]); }
                h2 pk = {(f16_t)r0, (f16_t)r1};
                *reinterpret_cast<unsigned*>(out + n) = as_u(pk);                   // N % 8 == 0: n+1 < N
            }
        }
    } else if constexpr (MODE == MODE_FFN) {
        float g, u;
        int n;
        if constexpr (COLS == 4) {        // row r: gate(col r) in g, up(col r) in u
            g = reduce4_q4(colsum[0][0], colsum[0][1], colsum[0][2], colsum[0][3]) * 1048576.f;
            u = reduce4_q4(colsum[NMAT - 1][0], colsum[NMAT - 1][1], colsum[NMAT - 1][2], colsum[NMAT - 1][3]) * 1048576.f;
            n = wg * 4 + row;
        } else {                // COLS == 2: rows = gate c0, up c0, gate c1, up c1; fetch the partner row
            const float w = reduce4_q4(colsum[0][0], colsum[NMAT - 1][0], colsum[0][1], colsum[NMAT - 1][1]) * 1048576.f;
            const float o = __shfl_xor(w, 16);
            g = (row & 1) ? o : w;
            u = (row & 1) ? w : o;
            n = wg * 2 + (row >> 1);
        }
        float val = g;
        val *= 1.0f / (1.0f + expf(-val));              // gpu_kernels.h:271
        val *= u;                                       // :272
        const bool wr = COLS == 4 ? writer : (lane & 31u) == 0;
        if (wr && n < N) a.out[0][n] = f2h(val);
    } else {   // MODE_QKV, COLS == 4: rows = pair0 first, pair1 first, pair0 second, pair1 second
        q4_half* out = a.out[mat0];
        const int pos = pos_now;
        if (mat0 != 0) out += (size_t)a.loff + (size_t)pos * N;                     // gpu_kernels.h:251,253
        const float mine = reduce4_q4(colsum[0][0], colsum[0][1], colsum[0][2], colsum[0][3]) * 1048576.f;
        const int hp = a.head_size >> 1;
        const int p = wg * 2 + (row & 1);                // pair index of this row
        const int head = p / hp, i = p - head * hp;
        const int n = head * a.head_size + i + ((row & 2) ? hp : 0);
        float r = mine;
        if (a.rope && mat0 < 2) {
            // RoPERotation_kernel gpu_kernels.h:332-355 on the fp16-rounded GEMV outputs
            const float other = round_h(__shfl_xor(mine, 32));
            const float me = round_h(mine);
            float fcr, fci;
            if (a.rope_table != nullptr) {   // same bits as the on-the-fly path: the table was filled by rope_angle()
                const float2 cs = a.rope_table[(size_t)pos * hp + i];
                fcr = cs.x; fci = cs.y;
            } else {
                rope_angle(i, a.head_size, pos, a.rope_theta, fcr, fci);
            }
            r = (row & 2) ? (other * fci + me * fcr) : (me * fcr - other * fci);    // :345-346
        }
        if (writer && n < N) out[n] = f2h(r);
        // the next launch of the stream (attention -> o-proj, layer_attn.h) tags its hand-off granules with this word: advanced
        // here, one launch ahead, so that every block of that launch reads the same value (stream order, not dispatch order).
        // At the very end: a memory operation in front of the weight loads costs the kernel its per-slot waits (+0.5 us measured)
        if (a.bump != nullptr && vbx == 0 && vby == 0 && tid == 0)
            __hip_atomic_fetch_add(a.bump, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ABL == 3 && a.dbg != nullptr && lane == 0) {
        ts[7] = __builtin_readcyclecounter();
        unsigned long long* d = a.dbg + (size_t)wg * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = ts[i];
    }
}

template <int MODE, int SLOTS, int COLS, bool NORM, int ABL = 0, int KS = 1, bool HALF = false>
__global__ void __launch_bounds__((LaunchTraits<MODE, SLOTS, COLS, KS>::MAX_THREADS)) gemv_q4_kernel(const GemvArgs a) {
    gemv_q4_body<MODE, SLOTS, COLS, NORM, ABL, KS, HALF, ROLE_NONE>(a, blockIdx.x, blockIdx.y, Handoff{});
}

// CUs of the CURRENT device (per ordinal: a process may drive several devices through q4_set_device)
static inline int cu_count() {
    static int n[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0 && (hipDeviceGetAttribute(&n[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n[dev] <= 0)) n[dev] = 256;
    return n[dev];
}

// host-side dispatch -------------------------------------------------------------------------------
extern int g_ablate;
extern int g_half_tail; // 1: shared half slot where the shape allows it (13B: K = 5120)
extern int g_ksplit;    // long-K plain GEMV: 2 waves per column group (default), 0: one
template <int MODE, int SLOTS, int COLS, bool NORM, int ABL = 0, int KS = 1, bool HALF = false>
static int launch_one(const GemvArgs& a0, int waves) {
    if (waves * 64 > LaunchTraits<MODE, SLOTS, COLS, KS>::MAX_THREADS) waves = LaunchTraits<MODE, SLOTS, COLS, KS>::MAX_THREADS / 64;
    waves -= waves % KS;
    const int cols_per_block = COLS * (waves / KS);
    dim3 grid(divUp(a0.N, cols_per_block), MODE == MODE_QKV ? 3 : 1);
    GemvArgs a = a0;
    if (a.early > 0 && a.early < 256) {
        // early birds hold back until the launch's x loads are queued: that takes longer the more waves share a CU
        // (measured, tools/lab/sweep_early.py: <= 12 waves per CU none, 15 -> 4 x 128 cycles, >= 21 -> 8 x 128)
        const int waves_per_cu = (int)((size_t)grid.x * grid.y * waves / (size_t)cu_count());
        a.early |= (waves_per_cu <= 12 ? 0 : waves_per_cu <= 16 ? 4 : 8) << 8;
    }
    const size_t smem = LdsLayout<SLOTS, KS>::BYTES;
    // long-K split kernels stage up to 32768 inputs: opt in to the CU's 160 KB (once per device)
    { const int rc = lds_opt_in((const void*)gemv_q4_kernel<MODE, SLOTS, COLS, NORM, ABL, KS, HALF>, smem); if (rc) return rc; }
    Q4_LAUNCH((gemv_q4_kernel<MODE, SLOTS, COLS, NORM, ABL, KS, HALF>), grid, dim3(waves * 64), smem, a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

// the last k-slot of a column holds at most 32 uint4: candidates for the shared half slot (K = 5120, 11008, ...)
static inline bool half_tail(const GemvArgs& a) { return g_half_tail && a.pw4 - (a.nslots - 1) * 64 <= 32; }

// smallest instantiated SLOTS covering nslots (K <= 16384), 0 if none
static inline int pick_slots(int nslots) {
    return nslots <= 2 ? 2 : nslots <= 3 ? 3 : nslots <= 4 ? 4 : nslots <= 6 ? 6 : nslots <= 7 ? 7 : nslots <= 8 ? 8 : 0;
}

}  // namespace q4
