// q4_kernels.hip -- the fp16 companions of the int4 GEMV, hand-written for gfx950 (wave64):
// rmsnorm, fp16 GEMV (classifier), RoPE, single-kernel attention, embedding gather, argmax,
// fp16->fp32 copy; plus the C-ABI launchers of include/llama2_q4.h for every kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include "gemv_strip_cls.h"   // (gemv_q4.h, the LDS-DMA helpers) + the classifier as strips
#include "attention.h"

namespace q4 {

int launch_gemv_plain(const GemvArgs& a, int cols, int waves);
int launch_gemv_qkv(const GemvArgs& a, int cols, int waves);
int launch_gemv_ffn(const GemvArgs& a, int cols, int waves);

// launch shapes (columns per wave, waves per block); tuned on MI355X, see DESIGN.md
enum { TUNE_PLAIN_SMALL = 0, TUNE_PLAIN_BIG = 1, TUNE_QKV = 2, TUNE_FFN = 3, TUNE_COUNT = 4 };
int g_ablate = 0;
int g_ksplit = 2;
int g_half_tail = 1;
int g_att_chunk = 0;         // positions per split-attention block (64 / 128 / 256), or 0 = measured per bin: the fused launch takes eight
                             // chunks per head (64 in bin 512, 128 in bin 1024, 256 above), the stand-alone kernels 128 up to bin 1024
int g_att_8wave = 0;         // head 128, bins 256 / 512: 8 waves x 8 loads in flight (1, the fused launch's shape) or 16 waves x 4
                            // (0: 8 us per token faster stand-alone, tools/lab/sweep_block2.py)
int g_att_split_min = 512;   // smallest sequence-length bin that uses the split-context kernels
unsigned long long* g_dbg = nullptr;
// early = 4: exactly the first block on each CU (measured: partial blocks or a second block lose the gain). The
// hold-back before the early loads (bits 8+, 128-cycle steps) is chosen per launch in launch_one(): with 21 waves per CU
// (gate/up) the x loads take ~0.5 us to queue and early weight requests in front of them make those blocks straggle
static GemvTune g_tune[TUNE_COUNT] = {{4, 4, 4}, {4, 4, 4}, {4, 4, 4}, {2, 4, 4}};

// ------------------------------------------------------------------------------------------------
// rmsnorm_kernel (gpu_kernels.h:72-105). One block; 16-byte loads; the canonical chunk-partial reduction
// of q4_device.h so that fused consumers reproduce these bits exactly.
__global__ void __launch_bounds__(256) rmsnorm_kernel(q4_half* o, const q4_half* x, const q4_half* weight, int size) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem);
    const int nchunks = size >> 3;
    for (int u = threadIdx.x; u < nchunks; u += blockDim.x)
        part[u] = sumsq8(reinterpret_cast<const u32x4*>(x)[u], 0.f);
    __syncthreads();
    const float ss = rms_scale_from_partials<0>(part, nchunks, size);
    for (int u = threadIdx.x; u < nchunks; u += blockDim.x)
        reinterpret_cast<u32x4*>(o)[u] =
            rms_apply8(reinterpret_cast<const u32x4*>(x)[u], reinterpret_cast<const u32x4*>(weight)[u], ss);
}

// ------------------------------------------------------------------------------------------------
// mat_vec_kernel (gpu_kernels.h:109-139): fp16 GEMV, used for the un-quantised classifier (262 MB at 7B).
// One wave owns ROWS consecutive output rows; lane l reads uint4 j = s*64 + l of each row (1 KiB per wave
// instruction, non-temporal); x lives in registers (same slice for every row); v_dot2c_f32_f16, fp32 acc.
template <int ROWS, int SLOTS>
__global__ void __launch_bounds__(256) gemv_f16_kernel(q4_half* op, const q4_half* ip, const q4_half* wt, int n, int d,
                                                       int ip_stride, int w_stride, int op_stride, int w_row_stride,
                                                       float alpha) {
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + wave) * ROWS;
    if (row0 >= d) return;
    const q4_half* input = ip + (size_t)blockIdx.y * ip_stride;
    const q4_half* weight = wt + (size_t)blockIdx.y * w_stride;
    q4_half* output = op + (size_t)blockIdx.y * op_stride;
    const unsigned n8 = (unsigned)n >> 3;
    float sum[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) sum[r] = 0.f;
    for (unsigned s0 = 0; s0 * 64 < n8; s0 += SLOTS) {
        u32x4 X[SLOTS], W[ROWS][SLOTS];
        bool act[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            const unsigned j = (s0 + s) * 64 + lane;
            act[s] = j < n8;
            const unsigned jj = act[s] ? j : n8 - 1;
            X[s] = reinterpret_cast<const u32x4*>(input)[jj];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int row = row0 + r < d ? row0 + r : d - 1;
                W[r][s] = ld_nt(reinterpret_cast<const u32x4*>(weight + (size_t)row * w_row_stride) + jj);
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; s++)
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_fdot2(as_h2(W[r][s][e]), as_h2(X[s][e]), acc, false);
                sum[r] += act[s] ? acc : 0.f;
            }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        float t = wave_sum(sum[r]);
        t *= alpha;                                                   // gpu_kernels.h:135
        if (lane == 0 && row0 + r < d) output[row0 + r] = f2h(t);
    }
}

// ------------------------------------------------------------------------------------------------
// RoPERotation_kernel (gpu_kernels.h:332-355), one block per head, head_size/2 threads.
__global__ void rope_kernel(q4_half* sq, q4_half* sk_base, int num_kv_heads, int head_size, const int* pPos, int loff,
                            float rope_theta) {
    const int pos = *pPos;
    const int h = blockIdx.x;
    q4_half* q = sq + (size_t)h * head_size;
    const int i = threadIdx.x;
    float fcr, fci;
    rope_angle(i, head_size, pos, rope_theta, fcr, fci);
    const float q0 = h2f(q[i]), q1 = h2f(q[i + head_size / 2]);
    q[i] = f2h(q0 * fcr - q1 * fci);
    q[i + head_size / 2] = f2h(q0 * fci + q1 * fcr);
    if (h < num_kv_heads) {
        q4_half* k = sk_base + (size_t)loff + (size_t)pos * num_kv_heads * head_size + (size_t)h * head_size;
        const float k0 = h2f(k[i]), k1 = h2f(k[i + head_size / 2]);
        k[i] = f2h(k0 * fcr - k1 * fci);
        k[i + head_size / 2] = f2h(k0 * fci + k1 * fcr);
    }
}

// ------------------------------------------------------------------------------------------------
// MultiHeadAttention: attention_body / attention_kernel live in attention.h (shared with the fused launch, layer_attn.hip)
// Long contexts: attention_split_body / attention_split_kernel / attention_combine_kernel live in attention.h

// ------------------------------------------------------------------------------------------------
// copy_embedding_kernel (gpu_kernels.h:61-69); 16-byte copies. tokens may live in mapped host memory.
__global__ void copy_embedding_kernel(q4_half* x, const q4_half* table, int size, const int* tokens, const int* pPos) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (size >> 3)) return;
    const int pos = *pPos;
    const int token = tokens[pos];
    reinterpret_cast<u32x4*>(x)[u] = reinterpret_cast<const u32x4*>(table + (size_t)token * size)[u];
}

// convert_fp16_to_fp32 (gpu_kernels.h:55-59). pos_stride != 0: out += *pPos * pos_stride (logits_array slot,
// llama2_q4.cu:380) so the launch is capturable in a graph.
__global__ void convert_fp16_to_fp32_kernel(float* out, const q4_half* in, int elements, const int* pPos,
                                            int pos_stride) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (pPos) out += (size_t)(*pPos) * pos_stride;
    if (index < elements) out[index] = h2f(in[index]);
}

// argmax_kernel (gpu_kernels.h:448-493). Ties resolve to the LOWEST index (the reference's tie-break is a
// benign race; lowest index is one of its legal outcomes and matches the oracle).
// x_next != nullptr (greedy steps queued several per graph replay, q4_run_transformer_steps): the winner's embedding row is
// copied into the residual stream right here -- the next step of the replay takes its token from this launch by construction, so
// its copy_embedding launch (a PCIe read of the pinned token ring + a boundary) is left out of the graph
__global__ void __launch_bounds__(1024) argmax_kernel(const q4_half* x, int size, int* result, volatile int* pPos,
                                                      int* pPosGpu, int write_token, q4_half* x_next, const q4_half* table, int dim) {
    __shared__ float sval[16];
    __shared__ int sidx[16];
    __shared__ int s_token;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the position lives in HBM too (pPosGpu == *pPos by construction, :490-491): read that copy instead of
    // paying a PCIe round trip to the pinned host word on every token
    int token_pos = *pPosGpu;
    float max_val = -INFINITY;
    int max_pos = 0x7fffffff;
    const int n8 = size >> 3;
    for (int u = tid; u < n8; u += blockDim.x) {              // 16-byte loads, first max wins inside a thread
        const u32x4 v = reinterpret_cast<const u32x4*>(x)[u];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const h2 p = as_h2(v[e]);
            const float a = (float)p.x, b = (float)p.y;
            if (a > max_val) { max_val = a; max_pos = u * 8 + 2 * e; }
            if (b > max_val) { max_val = b; max_pos = u * 8 + 2 * e + 1; }
        }
    }
    for (int i = n8 * 8 + tid; i < size; i += blockDim.x) {
        const float v = h2f(x[i]);
        if (v > max_val) { max_val = v; max_pos = i; }
    }
    // wave argmax by butterfly (value, then lower index)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(max_val, off);
        const int op = __shfl_xor(max_pos, off);
        if (ov > max_val || (ov == max_val && op < max_pos)) { max_val = ov; max_pos = op; }
    }
    if (lane == 0) { sval[wave] = max_val; sidx[wave] = max_pos; }
    __syncthreads();
    if (tid == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++)
            if (sval[w] > max_val || (sval[w] == max_val && sidx[w] < max_pos)) { max_val = sval[w]; max_pos = sidx[w]; }
        if (max_pos == 0x7fffffff) max_pos = 0;          // all NaN / -inf
        s_token = max_pos;
    }
    if (x_next != nullptr) {                             // (uniform: a kernel argument)
        __syncthreads();
        const int token = s_token;
        for (int u = tid; u < (dim >> 3); u += blockDim.x)
            reinterpret_cast<u32x4*>(x_next)[u] = reinterpret_cast<const u32x4*>(table + (size_t)token * dim)[u];
    }
    if (tid == 0) {
        token_pos++;
        if (write_token) result[token_pos] = max_pos;    // :486-487
        __threadfence_system();                          // the host may be spinning on *pPos (q4_wait_pos)
        *pPos = token_pos;                               // :490 (unblocks the CPU)
        *pPosGpu = token_pos;                            // :491
    }
}

// (cos, sin) of every (position, pair) with the reference's own formula, so that table lookups in the fused QKV
// epilogue are bit-identical to RoPERotation_kernel's on-the-fly powf/cosf/sinf (~300 VALU per wave saved)
__global__ void rope_table_kernel(float2* table, int seq_len, int head_size, float rope_theta) {
    const int hp = head_size >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if ((size_t)blockIdx.x * blockDim.x + threadIdx.x >= (size_t)seq_len * hp) return;
    float c, s;
    rope_angle(idx % hp, head_size, idx / hp, rope_theta, c, s);
    table[idx] = make_float2(c, s);
}

// One table per Transformer (owned by its slabs in q4_runtime.hip, freed with it): seq_len rows of head_size/2 pairs.
int rope_table_build(float2** out, int seq_len, int head_size, float theta) {
    *out = nullptr;
    const size_t n = (size_t)seq_len * (head_size / 2);
    if (n == 0 || n > (1u << 30)) return Q4_OK;   // no table: the kernels compute the angles
    if (hipMalloc((void**)out, n * sizeof(float2)) != hipSuccess) { *out = nullptr; (void)hipGetLastError(); return Q4_OK; }
    Q4_LAUNCH(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, *out, seq_len, head_size, theta);
    Q4_LAUNCH_CHECK();
    Q4_HIP(hipStreamSynchronize(g_stream));
    return Q4_OK;
}

// ---- fused-path launchers used by the network -----------------------------------------------------
static void fill_mat(GemvMat& m, const QWeight* w) { m.w = w->weight; m.z = w->zeros; m.s = w->scales; }

static int fill_geom(GemvArgs& a, int K, int N) {
    if ((K & 7) || (N & 7)) return Q4_ERR_UNSUPPORTED_SIZE;      // llama2_q4.cu:225
    if (K % 32) return Q4_ERR_UNSUPPORTED_SIZE;                   // packed height must be whole uint4
    QGeom g = make_geom(K, N);
    a.K = K; a.N = N; a.pw4 = g.pw4; a.pzh = g.pzh; a.sh = g.sh; a.nslots = g.nslots;
    if (g.nslots > 16) return Q4_ERR_UNSUPPORTED_SIZE;               // K > 32768: not instantiated (plain GEMV; fused kernels: K <= 16384)
    return Q4_OK;
}

int launch_qkv_fused(q4_half* q, q4_half* kc, q4_half* vc, const q4_half* x, const q4_half* rms_w, const QWeight* qw,
                     const QWeight* kw, const QWeight* vw, int dim, int kv_dim, long long loff, const int* pPos,
                     int head_size, float rope_theta, const float2* rope_table, unsigned* bump) {
    if (kv_dim > dim || (kv_dim & 7)) return Q4_ERR_ARG;
    GemvArgs a = {};
    int rc = fill_geom(a, dim, dim);
    if (rc) return rc;
    a.N_kv = kv_dim == dim ? 0 : kv_dim;      // grouped-query attention (llama2_q4.cu:309-313 runs three GEMVs there)
    fill_mat(a.m[0], qw); fill_mat(a.m[1], kw); fill_mat(a.m[2], vw);
    a.out[0] = q; a.out[1] = kc; a.out[2] = vc;
    a.x = x; a.rms_w = rms_w; a.pPos = pPos; a.loff = loff;
    a.rope = head_size > 0; a.head_size = head_size > 0 ? head_size : 2; a.rope_theta = rope_theta;
    a.rope_table = head_size > 0 ? rope_table : nullptr;   // [seq_len][head_size/2] for THIS model, or null: compute
    a.early = g_tune[TUNE_QKV].early;
    a.bump = bump;
    return launch_gemv_qkv(a, g_tune[TUNE_QKV].cols, g_tune[TUNE_QKV].waves);
}

int launch_ffn_fused(q4_half* out, const q4_half* x, const q4_half* rms_w, const QWeight* gate, const QWeight* up,
                     int dim, int hidden) {
    GemvArgs a = {};
    int rc = fill_geom(a, dim, hidden);
    if (rc) return rc;
    fill_mat(a.m[0], gate); fill_mat(a.m[1], up);
    a.out[0] = out; a.x = x; a.rms_w = rms_w; a.dbg = g_dbg;
    a.early = g_tune[TUNE_FFN].early;
    return launch_gemv_ffn(a, g_tune[TUNE_FFN].cols, g_tune[TUNE_FFN].waves);
}

}  // namespace q4

using namespace q4;

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

#ifdef Q4_PROFILING
// Measurement knobs of the profiling build (libllama2_q4_prof.so, used by tools/ and tests/prof_cases.py only): the
// shipped libllama2_q4.so exports exactly what include/llama2_q4.h declares.
void q4_set_ablate(int mode) { g_ablate = mode; }
void q4_set_ksplit(int on) { g_ksplit = on; }
void q4_set_half_tail(int on) { g_half_tail = on; q4_reset_graphs(); }
void q4_set_attention_split(int chunk, int min_bin) { g_att_chunk = chunk; g_att_split_min = min_bin; q4_reset_graphs(); }
void q4_set_gemv_early(int kind, int slots) {
    if (kind >= 0 && kind < TUNE_COUNT) g_tune[kind].early = slots;
    if (kind == 6) g_att_8wave = slots;
    if (kind == 7) g_multi_steps = slots;
    if (kind == 9) g_ao_mute = slots;       // the attention blocks of the next `slots` attention -> o-proj launches do not publish
    if (kind == 10) g_ao_vslice = slots;    // 0: one attention block per head below the split-context bins, 1: one per 64-byte V slice
    if (kind == 15) g_ao_hold_pct = slots;  // split-context bins: the o-proj role's weight requests held back this share of the K / V stream's estimated duration
    if (kind == 11) g_gemv_form = slots;    // a GemvForm (q4_internal.h): 0 = the product's choices, -1 = wave-owned kernels only, 1..6 loader / consumer engine, 8..19 strips settings
    if (kind == 20) g_ao16 = slots;         // attention -> o-proj below the split-context bins: 1 sixteen-wave blocks (layer_attn.h), 0 eight-wave
    if (kind == 16) g_fp_pre = slots;       // FFN pair launch (gemv_ffn_pair.h): down pieces requested in front of a wave's first gather pass
    if (kind == 17) g_fp_mute = slots;
    if (kind == 18) g_fp_nt = slots;      // ... the blocks of the next `slots` launches do not publish (a real time-out)
    if (kind == 8) g_ao_guard = slots;      // 0: admit attention -> o-proj grids beyond the resident capacity (forward-progress tests)
    q4_reset_graphs();
}
void q4_set_debug_buffer(void* p) { g_dbg = (unsigned long long*)p; }
void q4_set_gemv_tune(int kind, int cols, int waves) {
    if (kind >= 0 && kind < TUNE_COUNT && waves >= 4 && waves <= 8) { g_tune[kind].cols = cols; g_tune[kind].waves = waves; }
}
#endif

int q4_rmsnorm(q4_half* o, const q4_half* x, const q4_half* weight, int size) {
    if (size & 7) return Q4_ERR_UNSUPPORTED_SIZE;
    const size_t smem = (size_t)(size >> 3) * 4 + 16;
    Q4_LAUNCH(rmsnorm_kernel, dim3(1), dim3(256), smem, o, x, weight, size);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

int q4_matmul_f16(q4_half* xout, const q4_half* x, const q4_half* w, int n, int d, int batch, int x_stride, int w_stride,
                  int op_stride, int w_row_stride, float alpha) {
    if ((n & 7) || (d & 7)) return Q4_ERR_UNSUPPORTED_SIZE;                         // llama2_q4.cu:215
    if (w_row_stride == -1) w_row_stride = n;                                       // :220
    if (w_row_stride & 7) return Q4_ERR_UNSUPPORTED_SIZE;
    if (cls_strip_covers(n, d, batch, w_row_stride, alpha)) return launch_cls_strip(xout, x, nullptr, w, n, d);   // the classifier's shape: same bits, streamed through LDS-DMA rings
    constexpr int ROWS = 2, WAVES = 4;
    dim3 grid(divUp(d, ROWS * WAVES), batch);
    if (n <= 2048)
        Q4_LAUNCH((gemv_f16_kernel<ROWS, 4>), grid, dim3(WAVES * 64), 0, xout, x, w, n, d, x_stride,
                           w_stride, op_stride, w_row_stride, alpha);
    else
        Q4_LAUNCH((gemv_f16_kernel<ROWS, 8>), grid, dim3(WAVES * 64), 0, xout, x, w, n, d, x_stride,
                           w_stride, op_stride, w_row_stride, alpha);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

int q4_matmul_q4(q4_half* xout, const q4_half* x, const QWeight* w, int inpSize, int opSize, int accum, int loff,
                 const int* pPos) {
    GemvArgs a = {};
    int rc = fill_geom(a, inpSize, opSize);
    if (rc) return rc;
    if (loff != -1 && pPos == nullptr) return Q4_ERR_ARG;
    fill_mat(a.m[0], w);
    a.out[0] = xout; a.x = x; a.accum = accum; a.loff = loff; a.pPos = pPos;
    const GemvTune& t = g_tune[a.nslots <= 2 ? TUNE_PLAIN_SMALL : TUNE_PLAIN_BIG];
    a.early = t.early;
    return launch_gemv_plain(a, t.cols, t.waves);
}

int q4_qkv_matvec(q4_half* q, q4_half* key_cache, q4_half* value_cache, const q4_half* x, const QWeight* qw,
                  const QWeight* kw, const QWeight* vw, int inpSize, int opSize, int loff, const int* pPos) {
    GemvArgs a = {};
    int rc = fill_geom(a, inpSize, opSize);
    if (rc) return rc;
    if (pPos == nullptr) return Q4_ERR_ARG;
    fill_mat(a.m[0], qw); fill_mat(a.m[1], kw); fill_mat(a.m[2], vw);
    a.out[0] = q; a.out[1] = key_cache; a.out[2] = value_cache;
    a.x = x; a.pPos = pPos; a.loff = loff; a.rope = 0; a.head_size = 2;
    a.early = g_tune[TUNE_QKV].early;
    return launch_gemv_qkv(a, g_tune[TUNE_QKV].cols, g_tune[TUNE_QKV].waves);
}

int q4_ffn_matvec_silu(q4_half* xout, const q4_half* x, const QWeight* gate_w, const QWeight* up_w, int inpSize,
                       int opSize) {
    return launch_ffn_fused(xout, x, nullptr, gate_w, up_w, inpSize, opSize);
}

int q4_rope_rotation(q4_half* q, q4_half* k, int num_heads, int num_kv_heads, int head_size, const int* pPos, int loff,
                     float rope_theta) {
    if (head_size < 2 || head_size / 2 > 1024) return Q4_ERR_UNSUPPORTED_SIZE;
    Q4_LAUNCH(rope_kernel, dim3(num_heads), dim3(head_size / 2), 0, q, k, num_kv_heads, head_size, pPos,
                       loff, rope_theta);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // extern "C"

namespace q4 {
int launch_attention(q4_half* output, const q4_half* q, const q4_half* key_cache, const q4_half* value_cache,
                     int num_heads, int head_size, int kv_mul, int max_seq_len, const int* pPos, float* scratch,
                     size_t scratch_bytes, unsigned* arrive) {
    const int dim = head_size * num_heads;
    const int kv_dim = dim / kv_mul;
    const float alpha = (float)(1.0 / sqrt((double)head_size));                     // llama2_q4.cu:273
    dim3 block(ATT_NW * 64);
    // long context: one block per (head, 256-position chunk); merged by each head's last block (arrive != nullptr: the
    // model's counters) or by a second launch (see attention_split_kernel)
    const int want = g_att_chunk ? g_att_chunk : (max_seq_len <= 1024 ? 128 : 256);
    const int chunk = want == 128 && head_size == 128 ? 128 : 256;
    const int nsp = divUp(max_seq_len, chunk);
    const bool split = max_seq_len >= g_att_split_min && scratch != nullptr && (head_size == 64 || head_size == 128 || head_size == 256) &&
                       (size_t)num_heads * nsp * (head_size + ATT_REC_PAD) * sizeof(float) <= scratch_bytes;
    if (split) {
        const size_t smem = (size_t)(32 + ATT_NW * head_size) * 4;
        const dim3 grid(num_heads, nsp);
        const SplitArgs sa = {scratch, q, key_cache, value_cache, head_size, kv_mul, kv_dim, pPos, alpha, output, arrive};
#define Q4_SPLIT(L, UU) Q4_LAUNCH((attention_split_kernel<L, UU>), grid, block, smem, sa)
        if (head_size == 64) Q4_SPLIT(8, 2);                 // 16 waves x 8 rows x 2 = 256 positions per block
        else if (head_size == 256) Q4_SPLIT(32, 8);          // 16 x 2 x 8
        else if (chunk == 128) Q4_SPLIT(16, 2);
        else Q4_SPLIT(16, 4);
#undef Q4_SPLIT
        if (arrive == nullptr)
            Q4_LAUNCH(attention_combine_kernel, dim3(num_heads), dim3(128), 0, output, (const float*)scratch, head_size, nsp);
        Q4_LAUNCH_CHECK();
        return Q4_OK;
    }
    // LDS: 32 reduction floats + NW*head_size output partials + one fp32 score per position
    const size_t smem = (size_t)(32 + ATT_NW * head_size + max_seq_len) * 4;
    if (smem > 160 * 1024) return Q4_ERR_UNSUPPORTED_SIZE;                          // > ~38K positions
    dim3 grid(num_heads);
    const AttArgs aa = {output, q, key_cache, value_cache, head_size, kv_mul, kv_dim, pPos, alpha, max_seq_len, g_dbg};
#define Q4_ATT(L)                                                                                                  \
    {                                                                                                              \
        { const int rc = lds_opt_in((const void*)attention_kernel<L>, smem); if (rc) return rc; }   /* only when a launch needs more than 64 KiB */ \
        Q4_LAUNCH((attention_kernel<L>), grid, block, smem, aa);                                                   \
    }
    switch (head_size) {
        case 32: Q4_ATT(4) break;
        case 64: Q4_ATT(8) break;
        case 128:
            if (max_seq_len <= 128) {   // first bin: 8 waves cover the 128 positions in one pass (no clamped duplicate loads,
                                        // cheaper barriers): 7B -n 256 +1 % over the 16-wave block
                Q4_LAUNCH((attention_kernel<16, 4, 8>), grid, dim3(8 * 64), smem, aa);
            } else if (max_seq_len <= 512 && g_att_8wave) {   // the shape of the fused launch's attention role (same bits)
                Q4_LAUNCH((attention_kernel<16, 8, 8>), grid, dim3(8 * 64), smem, aa);
            } else Q4_ATT(16)
            break;
        case 256: Q4_ATT(32) break;
        default: {
            // any other multiple of 8 up to 256 (the reference's kernels take any head size): the next wider instantiation with the
            // lanes past the head masked
            if (head_size < 8 || head_size > 256 || (head_size & 7)) return Q4_ERR_UNSUPPORTED_SIZE;
#define Q4_ATT_PAD(L)                                                                                                       \
    {                                                                                                                       \
        { const int rc = lds_opt_in((const void*)attention_kernel<L, 4, ATT_NW, true>, smem); if (rc) return rc; }         \
        Q4_LAUNCH((attention_kernel<L, 4, ATT_NW, true>), grid, block, smem, aa);                                           \
    }
            if (head_size < 32) Q4_ATT_PAD(4) else if (head_size < 64) Q4_ATT_PAD(8) else if (head_size < 128) Q4_ATT_PAD(16) else Q4_ATT_PAD(32)
#undef Q4_ATT_PAD
        }
    }
#undef Q4_ATT
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
}  // namespace q4

extern "C" {

int q4_multi_head_attention(q4_half* output, const q4_half* q, const q4_half* key_cache, const q4_half* value_cache,
                            q4_half* att, int num_heads, int head_size, int kv_mul, int max_seq_len, const int* pPos) {
    // `att` (n_heads * max_seq_len halves in the reference) doubles as the split-context scratch when it is big enough
    const size_t att_bytes = att ? (size_t)num_heads * max_seq_len * sizeof(q4_half) : 0;
    return launch_attention(output, q, key_cache, value_cache, num_heads, head_size, kv_mul, max_seq_len, pPos, (float*)att,
                            att_bytes, nullptr);       // the caller owns `att`: no counters in it, merged by a second launch
}

int q4_copy_embedding(q4_half* x, const q4_half* table, int size, const int* tokens, const int* pPos) {
    if (size & 7) return Q4_ERR_UNSUPPORTED_SIZE;
    Q4_LAUNCH(copy_embedding_kernel, dim3(divUp(size >> 3, 256)), dim3(256), 0, x, table, size, tokens,
                       pPos);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

int q4_convert_fp16_to_fp32(float* out, const q4_half* in, int elements) {
    Q4_LAUNCH(convert_fp16_to_fp32_kernel, dim3(divUp(elements, 256)), dim3(256), 0, out, in, elements,
                       (const int*)nullptr, 0);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

// logits -> logits_array[*pPos] (llama2_q4.cu:377-382), position read on the device so it can be graph-captured
__attribute__((visibility("hidden"))) int q4_copy_logits_at_pos(float* logits_array, const q4_half* logits, int vocab_size, const int* pPos) {
    Q4_LAUNCH(convert_fp16_to_fp32_kernel, dim3(divUp(vocab_size, 256)), dim3(256), 0, logits_array,
                       logits, vocab_size, pPos, vocab_size);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

int q4_argmax(const q4_half* x, int size, int* result, volatile int* pPos, int* pPosGpu, int write_token) {
    Q4_LAUNCH(argmax_kernel, dim3(1), dim3(1024), 0, x, size, result, pPos, pPosGpu, write_token, (q4_half*)nullptr, (const q4_half*)nullptr, 0);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // extern "C"

namespace q4 {
// argmax whose winner's embedding row becomes the next step's residual stream (see argmax_kernel)
int launch_argmax_feed(const q4_half* x, int size, int* result, volatile int* pPos, int* pPosGpu, q4_half* x_next, const q4_half* table, int dim) {
    if (dim & 7) return Q4_ERR_UNSUPPORTED_SIZE;
    Q4_LAUNCH(argmax_kernel, dim3(1), dim3(1024), 0, x, size, result, pPos, pPosGpu, 1, x_next, table, dim);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
}  // namespace q4

// final rmsnorm + classifier (llama2_q4.cu:336, 339) as ONE launch where the strips form covers the shape: the norm is computed once per CU inside it and x
// itself is left un-normalised (nothing reads it afterwards: the next step's embedding overwrites it); elsewhere the two launches of the reference
namespace q4 {
// the LDS opt-in is not a stream operation: q4_set_device and build_transformer make it (q4_runtime.hip), outside any capture
int cls_strip_prepare() {
    // called in front of every launch of the kernel: one hipGetDevice and a comparison once the device's opt-ins are made (single host thread: q4_internal.h)
    static int prepared_device = -1;
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev == prepared_device) return Q4_OK;
    int rc = Q4_OK;     // (once per device: lds_opt_in)
    if (!rc) rc = lds_opt_in((const void*)cls_strip_kernel<8, true, CLS_D>, StripClsLds<8, CLS_D>::BYTES);
    if (!rc) rc = lds_opt_in((const void*)cls_strip_kernel<8, false, CLS_D>, StripClsLds<8, CLS_D>::BYTES);
    if (!rc) rc = lds_opt_in((const void*)cls_strip_kernel<10, true, CLS_D>, StripClsLds<10, CLS_D>::BYTES);
    if (!rc) rc = lds_opt_in((const void*)cls_strip_kernel<10, false, CLS_D>, StripClsLds<10, CLS_D>::BYTES);
    if (!rc) prepared_device = dev;
    return rc;
}
// (the greedy sampler as this launch's epilogue was built and measured level in round 4: EXPERIMENTS.md #19)
int classifier_with_final_norm(q4_half* logits, q4_half* x, const q4_half* rms_w, const q4_half* wcls, int dim, int vocab) {
    if (cls_strip_covers(dim, vocab, 1, dim, 1.0f)) return launch_cls_strip(logits, x, rms_w, wcls, dim, vocab);
    const int rc = q4_rmsnorm(x, x, rms_w, dim);
    return rc ? rc : q4_matmul_f16(logits, x, wcls, dim, vocab, 1, 0, 0, 0, -1, 1.0f);
}
}  // namespace q4
