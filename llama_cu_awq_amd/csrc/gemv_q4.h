// gemv_q4.h -- the INT4-AWQ dequant-fused GEMV for gfx950 (wave64), all three launch shapes of the
// reference share this one kernel template:
//   MODE_PLAIN : mat_vec_kernel_int4      gpu_kernels.h:213-240  (accum / KV addressing)
//   MODE_QKV   : qkv_matvec_kernel        gpu_kernels.h:242-254  (+ optional RoPE epilogue :332-355)
//   MODE_FFN   : ffn_matvec_silu_kernel   gpu_kernels.h:256-275
// and optionally fold rmsnorm (gpu_kernels.h:72-105) into the activation staging.
//
// Mapping (designed from the data layout, SURVEY a1, not from the reference's 32x4 blocks):
//  * a column n is a contiguous run of K/32 uint4 (32 nibbles each). One WAVE owns COLS columns; lane l
//    reads uint4 j = slot*64 + l of each of them with one 128-bit non-temporal load -> a wave instruction
//    moves 1 KiB of one column, fully coalesced. All SLOTS x COLS (x 2 for gate/up) loads of a wave are
//    issued before anything is consumed, so the whole matrix is in flight at once (q/k/v/o are only
//    ~32 KiB per CU: one HBM round trip, no second wave of requests).
//  * the activation vector is staged once per block into LDS, pre-permuted so that the 16 B a lane needs
//    for weight dword d are one conflict-free ds_read_b128: unit [(slot*4 + d)*64 + lane] holds
//    (x0,x4),(x1,x5),(x2,x6),(x3,x7) of the 8 inputs that dword multiplies.
//  * dequant without int->float converts: (w & 0x000F000F) | 0x64006400 is the half2 (1024+q_i, 1024+q_{i+4});
//    v_pk_add_f16 with -(1024+z) gives the exact (q-z) pair; the odd nibbles sit 4 bits higher, i.e.
//    (1024+16q), and v_pk_fma_f16 by 1/16 with -(64+z) gives exact (q-z) again. v_dot2c_f32_f16 then
//    accumulates (q-z)*x in fp32; the group scale is applied once per 32 weights. 13 VALU per 8 weights.
//  * wave reduction by DPP row ops + 4 v_readlane (no LDS, no cub).
// No MFMA: 3.84 flop/B, the kernel is an HBM stream.
#pragma once
#include "q4_device.h"
#include "q4_internal.h"

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
    int pw4, pzh, sh, nslots;
    int accum;              // PLAIN: out = half(float(out) + sum)
    int loff;               // PLAIN: -1 = none. QKV: KV-cache layer offset
    int rope;               // QKV: rotate q and k in the epilogue
    int head_size;
    float rope_theta;
};

template <int MODE>
struct ModeTraits { static constexpr int NMAT = (MODE == MODE_FFN) ? 2 : 1; };

// one 32-weight unit: w = packed nibbles (uint4), xs = 4 permuted activation units, returns sum (q-z)*x
__device__ __forceinline__ float dot32_q4(u32x4 w, const u32x4 (&X)[4], unsigned z, unsigned M0, unsigned M1,
                                          unsigned MG) {
    const h2 cz0 = as_h2(0xE400E400u + z * 0x00010001u);   // -(1024+z) x2
    const h2 cz1 = as_h2(0xD400D400u + z * 0x00100010u);   // -(64+z)   x2
    const h2 s16 = {(f16_t)0.0625f, (f16_t)0.0625f};
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const unsigned ww = w[d];
        const unsigned t = ww >> 8;
        const h2 q0 = as_h2((ww & M0) | MG) + cz0;            // (n0, n4)
        const h2 q1 = as_h2((ww & M1) | MG) * s16 + cz1;      // (n1, n5)
        const h2 q2 = as_h2((t & M0) | MG) + cz0;             // (n2, n6)
        const h2 q3 = as_h2((t & M1) | MG) * s16 + cz1;       // (n3, n7)
        acc = __builtin_amdgcn_fdot2(q0, as_h2(X[d][0]), acc, false);
        acc = __builtin_amdgcn_fdot2(q1, as_h2(X[d][1]), acc, false);
        acc = __builtin_amdgcn_fdot2(q2, as_h2(X[d][2]), acc, false);
        acc = __builtin_amdgcn_fdot2(q3, as_h2(X[d][3]), acc, false);
    }
    return acc;
}

// permute 8 consecutive halves (x0..x7 as 4 dwords) into (x0,x4),(x1,x5),(x2,x6),(x3,x7)
__device__ __forceinline__ u32x4 permute_x8(u32x4 v) {
    u32x4 o;
    o[0] = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u);   // lo16(v0) | lo16(v2)<<16
    o[1] = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);   // hi16(v0) | hi16(v2)<<16
    o[2] = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u);
    o[3] = __builtin_amdgcn_perm(v[3], v[1], 0x07060302u);
    return o;
}

template <int MODE, int SLOTS, int COLS, bool NORM>
__global__ void __launch_bounds__(512) gemv_q4_kernel(const GemvArgs a) {
    constexpr int NMAT = ModeTraits<MODE>::NMAT;
    constexpr int NUNITS = SLOTS * 256;         // 16-byte LDS units (zero padded past K)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                          // [SLOTS][4][64]
    float* part = reinterpret_cast<float*>(smem + (size_t)NUNITS * 16);  // [NUNITS] chunk partials + bcast

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: column bases stay in SGPRs
    const int nw = blockDim.x >> 6;
    const int wg = blockIdx.x * nw + wave;      // global wave index
    const int mat0 = (MODE == MODE_QKV) ? blockIdx.y : 0;

    // ---- column ownership ---------------------------------------------------------------------
    int col[COLS];
    if (MODE == MODE_QKV) {
        // pairs (i, i + head_size/2) of one head live in the same wave so RoPE needs no exchange
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
    bool valid[COLS];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        valid[c] = col[c] < a.N;
        col[c] = valid[c] ? col[c] : a.N - 1;   // clamp: loads stay in bounds, the store is skipped
    }

    // ---- 1. activation loads first: their wait (counted vmcnt) leaves the weight loads in flight ----
    const unsigned nchunks = (unsigned)a.K >> 3;            // real 8-half chunks
    u32x4 xraw[SLOTS], wraw[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
        const unsigned u = tid + i * blockDim.x;
        const unsigned uc = u < nchunks ? u : nchunks - 1;  // clamped, branch-free
        xraw[i] = reinterpret_cast<const u32x4*>(a.x)[uc];
        if (NORM) wraw[i] = reinterpret_cast<const u32x4*>(a.rms_w)[uc];
    }

    float old[COLS];
    if (MODE == MODE_PLAIN) {
        const q4_half* ob = a.out[0];
        if (a.loff != -1) ob += (size_t)a.loff + (size_t)(*a.pPos) * a.N;
#pragma unroll
        for (int c = 0; c < COLS; c++) old[c] = a.accum ? h2f(ob[col[c]]) : 0.f;
    }

    // ---- 2. issue every weight / zero / scale load of this wave ---------------------------------
    u32x4 W[NMAT][SLOTS][COLS];
    unsigned ZW[NMAT][SLOTS][COLS];
    uint16_t SC[NMAT][SLOTS][COLS];
    bool act[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        const unsigned j = s * 64 + lane;
        act[s] = j < (unsigned)a.pw4;
        const unsigned jj = act[s] ? j : (unsigned)a.pw4 - 1;
#pragma unroll
        for (int m = 0; m < NMAT; m++) {
            const GemvMat& M = a.m[mat0 + m];
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                const uint32_t* zc = M.z + (size_t)col[c] * a.pzh;       // wave-uniform bases
                const q4_half* sc = M.s + (size_t)col[c] * a.sh;
                const u32x4* wc = reinterpret_cast<const u32x4*>(M.w) + (size_t)col[c] * a.pw4;
                ZW[m][s][c] = zc[jj >> 5];
                SC[m][s][c] = sc[jj >> 2];
                W[m][s][c] = ld_nt(wc + jj);
            }
        }
    }

    // ---- 3. stage the activation vector into LDS, optional fused rmsnorm -----------------------
    {
        float ss = 1.f;
        if (NORM) {
#pragma unroll
            for (int i = 0; i < SLOTS; i++) {
                const unsigned u = tid + i * blockDim.x;
                if (u < NUNITS) part[u] = u < nchunks ? sumsq8(xraw[i], 0.f) : 0.f;
            }
            __syncthreads();
            ss = rms_scale_from_partials(part, NUNITS, a.K, part + NUNITS);
        }
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            const unsigned u = tid + i * blockDim.x;
            u32x4 v = xraw[i];
            if (NORM) v = rms_apply8(v, wraw[i], ss);
            if (u >= nchunks) v = (u32x4){0u, 0u, 0u, 0u};
            const unsigned j = u >> 2, d = u & 3u;
            if (u < NUNITS) xs[(((j >> 6) * 4 + d) << 6) + (j & 63u)] = permute_x8(v);
        }
        __syncthreads();
    }

    unsigned M0 = 0x000F000Fu, M1 = 0x00F000F0u, MG = 0x64006400u;
    asm volatile("" : "+v"(M0), "+v"(M1), "+v"(MG));   // keep in VGPRs so (w & M) | MG selects v_and_or_b32

    // ---- 4. consume in issue order -------------------------------------------------------------
    float colsum[NMAT][COLS];
#pragma unroll
    for (int m = 0; m < NMAT; m++)
#pragma unroll
        for (int c = 0; c < COLS; c++) colsum[m][c] = 0.f;

#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        u32x4 X[4];
#pragma unroll
        for (int d = 0; d < 4; d++) X[d] = xs[((s * 4 + d) << 6) + lane];
        const unsigned j = s * 64 + lane;
        const unsigned zsh = ((j >> 2) & 7u) * 4u;
#pragma unroll
        for (int m = 0; m < NMAT; m++)
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                const unsigned z = (ZW[m][s][c] >> zsh) & 0xFu;
                const float scale = act[s] ? h2f(SC[m][s][c]) : 0.f;
                const float acc = dot32_q4(W[m][s][c], X, z, M0, M1, MG);
                colsum[m][c] = __builtin_fmaf(scale, acc, colsum[m][c]);
            }
    }

    // ---- 5. wave reduction + epilogue ----------------------------------------------------------
#pragma unroll
    for (int m = 0; m < NMAT; m++)
#pragma unroll
        for (int c = 0; c < COLS; c++) colsum[m][c] = wave_sum(colsum[m][c]);

    if (MODE == MODE_PLAIN) {
        q4_half* out = a.out[0];
        if (a.loff != -1) out += (size_t)a.loff + (size_t)(*a.pPos) * a.N;         // gpu_kernels.h:225-227
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; c++)
                if (valid[c]) out[col[c]] = f2h(a.accum ? colsum[0][c] + old[c] : colsum[0][c]);   // :229-231
        }
    } else if (MODE == MODE_FFN) {
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                float val = colsum[0][c];                       // gate
                val *= 1.0f / (1.0f + expf(-val));              // gpu_kernels.h:271
                val *= colsum[1][c];                            // up, :272
                if (valid[c]) a.out[0][col[c]] = f2h(val);
            }
        }
    } else {   // MODE_QKV
        constexpr int P = COLS / 2;
        q4_half* out = a.out[mat0];
        int pos = 0;
        if (mat0 != 0 || a.rope) pos = *a.pPos;
        if (mat0 != 0) out += (size_t)a.loff + (size_t)pos * a.N;                   // gpu_kernels.h:251,253
        float r[COLS];
#pragma unroll
        for (int c = 0; c < COLS; c++) r[c] = colsum[0][c];
        if (a.rope && mat0 < 2) {
            // RoPERotation_kernel gpu_kernels.h:332-355 on the fp16-rounded GEMV outputs
#pragma unroll
            for (int pi = 0; pi < P; pi++) {
                const int i = col[pi] % a.head_size;
                const int head_dim = (i * 2) % a.head_size;
                const float freq = 1.0f / powf(a.rope_theta, head_dim / (float)a.head_size);
                const float val = pos * freq;
                const float fcr = cosf(val), fci = sinf(val);
                const float v0 = round_h(r[pi]), v1 = round_h(r[P + pi]);
                r[pi] = v0 * fcr - v1 * fci;
                r[P + pi] = v0 * fci + v1 * fcr;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; c++)
                if (valid[c]) out[col[c]] = f2h(r[c]);
        }
    }
}

// host-side dispatch -------------------------------------------------------------------------------
template <int MODE, int SLOTS, int COLS, bool NORM>
static int launch_one(const GemvArgs& a, int waves) {
    const int cols_per_block = COLS * waves;
    dim3 grid(divUp(a.N, cols_per_block), MODE == MODE_QKV ? 3 : 1);
    const size_t smem = (size_t)SLOTS * 256 * 16 + (size_t)SLOTS * 256 * 4 + 16;
    Q4_LAUNCH((gemv_q4_kernel<MODE, SLOTS, COLS, NORM>), grid, dim3(waves * 64), smem, a);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

// smallest instantiated SLOTS covering nslots (K <= 16384), 0 if none
static inline int pick_slots(int nslots) {
    return nslots <= 2 ? 2 : nslots <= 3 ? 3 : nslots <= 4 ? 4 : nslots <= 6 ? 6 : nslots <= 7 ? 7 : nslots <= 8 ? 8 : 0;
}

}  // namespace q4
