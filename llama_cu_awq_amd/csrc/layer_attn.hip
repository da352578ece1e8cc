// layer_attn.hip -- the attention block of a layer as ONE launch: fused rmsnorm + q/k/v GEMV + RoPE + KV write
// (llama2_q4.cu:300-317), MultiHeadAttention (:320) and the output projection with residual add (:323).
//
// Every launch of the decode step carries ~3-4 us that are not streaming (boundary, kernel arguments, first-data latency,
// the x chain). Three launches become one: blocks [0, nq) are the QKV GEMV (producers), blocks [nq, nq + heads) the
// attention heads, the rest the o-proj GEMV. Work-groups are dispatched in index order, so every producer is resident
// before the first consumer; consumers only ever wait on blocks with LOWER indices, which wait on nobody -- forward
// progress does not depend on co-residency. While the QKV blocks stream their 26 MB the attention blocks already hold the
// cached K/V rows of positions < pos in registers and the o-proj blocks their 8.7 MB of weights; after the hand-offs only
// the dependent part is left: q + one cache row + softmax, then 8 KB of x + the o-proj dot products.
// Hand-off forms: MI355X_MICROARCH.md "Valid forms" (sc1 payload stores -> s_waitcnt vmcnt(0) -> relaxed agent-scope
// arrival; consumer: ONE lane polls relaxed with s_sleep, bounded, then the block reads the payload with sc1 loads).
// The arithmetic is the stand-alone kernels' own (gemv_q4_body, attention_body): results are bit-identical to the
// five-launch sequence (tests/test_forward_gpu.py::test_fused_equals_unfused_bits).
#include <hip/hip_runtime.h>
#include "attention.h"

namespace q4 {

constexpr int LA_WAVES = 8;          // 512-thread blocks for every role
int g_la_early = 0;                  // early-bird waves of the QKV role (profiling build: q4_set_gemv_early(4, early))

struct AttBlockArgs {
    GemvArgs qkv;
    GemvArgs oproj;
    AttArgs att;
    unsigned* sync;                  // hand-off words (sync_layout below), all zero between launches
    unsigned nq, nqx, nheads, no;    // QKV blocks (nqx per matrix), attention blocks, o-proj blocks
    unsigned long long* dbg;         // profiling build: [block][4] wall-clock stamps (entry, after the wait, end, role)
};

template <int SLOTS, bool HALF, int U>
__global__ void __launch_bounds__(LA_WAVES * 64) attention_block_kernel(const AttBlockArgs a) {
    const unsigned b = blockIdx.x;
    // sync layout (words): [0] sticky error flag, [32] epoch; granule vectors (8 bytes each) behind the first 4 KB:
    // q, k, v of this position ([3][dim/2]), then the attention output ([dim/2])
    Handoff ho = {};
    ho.error = a.sync;
    unsigned* epoch = a.sync + 32;
    ho.tag = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    u32x2v* g_qkv = reinterpret_cast<u32x2v*>(a.sync + 1024);
    u32x2v* g_att = g_qkv + 3 * (size_t)(a.qkv.N / 2);
#ifdef Q4_PROFILING
    if (a.dbg && threadIdx.x == 0) { a.dbg[b * 4 + 0] = wall_clock64(); a.dbg[b * 4 + 3] = (b < a.nq ? 0 : b < a.nq + a.nheads ? 1 : 2) | ((unsigned long long)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4) << 8) |
                                                                ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) << 32); }
    ho.stamp = a.dbg ? a.dbg + b * 4 + 1 : nullptr;
#endif
    if (b < a.nq) {
        ho.pub = g_qkv;
        gemv_q4_body<MODE_QKV, SLOTS, 4, true, 0, 1, HALF, ROLE_PRODUCER>(a.qkv, b % a.nqx, b / a.nqx, ho);
    } else if (b < a.nq + a.nheads) {
        ho.sub = g_qkv;
        ho.pub = g_att;
        attention_body<16, U, LA_WAVES, 1>(a.att, (int)(b - a.nq), ho);
    } else {
        const unsigned j = b - a.nq - a.nheads;
        ho.sub = g_att;
        ho.sentinel = (int)((j % a.nheads) * (a.att.head_size / 2) + a.att.head_size / 2 - 1);   // last granule of head j % heads
        gemv_q4_body<MODE_PLAIN, SLOTS, 4, false, 5, 1, HALF, ROLE_CONSUMER>(a.oproj, j, 0, ho);
        // the launch's last block (dispatched last: every block has read the epoch long before it ends) opens the next epoch
        if (b == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_fetch_add(epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef Q4_PROFILING
    if (a.dbg && threadIdx.x == 0) a.dbg[b * 4 + 2] = wall_clock64();
#endif
}

// attention -> o-proj as one launch (fusion level 3): the QKV GEMV keeps its own launch in its tuned 4-wave shape; here the
// attention runs in 8-wave blocks -- ATT 0: one block per head, 4 rows in flight per lane (first bin); 1: 8 rows (bin 256);
// 2 / 3: one block per (head, 128 / 256 positions), merged by each head's last block (bins >= 512) -- and the o-proj blocks
// (8 waves as well) pull their weights while the heads work.
struct AttOprojArgs {
    GemvArgs oproj;
    AttArgs att;
    SplitArgs split;
    unsigned* sync;
    unsigned nheads, natt, no;       // heads, attention blocks (heads, or heads x chunks), o-proj blocks
    unsigned long long* dbg;
};

template <int SLOTS, bool HALF, int ATT>
__global__ void __launch_bounds__(LA_WAVES * 64) attention_oproj_kernel(const AttOprojArgs a) {
    constexpr int NW = LA_WAVES;       // 8-wave blocks for both roles: 16-wave o-proj blocks (64 of them) lost 13-20 us per token
    const unsigned b = blockIdx.x;
    Handoff ho = {};
    ho.error = a.sync;
    unsigned* epoch = a.sync + 64;                      // this launch's own epoch word
    ho.tag = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    u32x2v* g_att = reinterpret_cast<u32x2v*>(a.sync + 1024) + 3 * (size_t)(a.oproj.N / 2);
#ifdef Q4_PROFILING
    if (a.dbg && threadIdx.x == 0) { a.dbg[b * 4 + 0] = wall_clock64(); a.dbg[b * 4 + 3] = b < a.natt ? 1 : 2; }
    ho.stamp = a.dbg ? a.dbg + b * 4 + 1 : nullptr;
#endif
    if (b < a.natt) {
        ho.pub = g_att;
        if constexpr (ATT <= 1) attention_body<16, ATT == 0 ? 4 : 8, NW, 2>(a.att, (int)b, ho);
        else attention_split_body<16, ATT == 2 ? 4 : 8, true, NW>(a.split, (int)(b % a.nheads), (int)(b / a.nheads), (int)(a.natt / a.nheads), ho);
    } else {
        const unsigned j = b - a.natt;
        ho.sub = g_att;
        ho.sentinel = (int)((j % a.nheads) * (a.att.head_size / 2) + a.att.head_size / 2 - 1);
        gemv_q4_body<MODE_PLAIN, SLOTS, 4, false, 5, 1, HALF, ROLE_CONSUMER>(a.oproj, j, 0, ho);
        if (b == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_fetch_add(epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef Q4_PROFILING
    if (a.dbg && threadIdx.x == 0) a.dbg[b * 4 + 2] = wall_clock64();
#endif
}

static void fill_mat(GemvMat& m, const QWeight* w) { m.w = w->weight; m.z = w->zeros; m.s = w->scales; }

// geometry that has this form: multi-head, head 128, K = dim in 2 slots or 3 with a shared half slot; any bin the attention
// kernels cover (the split form needs the scratch)
// Which attention form the launch would use for this geometry and bin: 0 / 1 one block per head (4 / 8 rows in flight),
// 2 / 3 split context (128 / 256 positions per block), or -1 when there is no fused form (the caller then runs the
// stand-alone launches): multi-head or grouped-query, head 128, K = dim in 2 slots or 3 with a shared half slot, and the
// one-block form only while its score buffer fits the default 64 KB of LDS.
int attention_oproj_form(int dim, int kv_dim, int head_size, int n_heads, int seq_len_bin, bool have_scratch, size_t scratch_bytes,
                         int split_min, int split_chunk) {
    const QGeom g = make_geom(dim, dim);
    const bool slots_ok = g.nslots == 2 || (g.nslots == 3 && g.pw4 - 2 * 64 <= 32);
    if (!(kv_dim > 0 && dim % kv_dim == 0 && head_size == 128 && slots_ok && (dim % 64) == 0)) return -1;
    const int chunk = split_chunk ? split_chunk : (seq_len_bin <= 512 ? 128 : 256);
    const int nsp = divUp(seq_len_bin, chunk);
    const bool split = seq_len_bin >= split_min && have_scratch && n_heads <= 512 &&
                       (size_t)n_heads * nsp * (head_size + ATT_REC_PAD) * sizeof(float) <= scratch_bytes;
    if (split) return chunk == 128 ? 2 : 3;
    if ((size_t)(32 + LA_WAVES * head_size + seq_len_bin) * 4 > 64 * 1024) return -1;
    return seq_len_bin <= 128 ? 0 : 1;
}

int launch_attention_oproj(q4_half* x, q4_half* xb, const q4_half* q, const q4_half* key_cache, const q4_half* value_cache,
                           const QWeight* wo, int dim, int kv_dim, int n_heads, const int* pPos, int seq_len_bin, unsigned* sync,
                           float* scratch, size_t scratch_bytes, int split_min, int split_chunk) {
    const int head_size = dim / n_heads;
    const QGeom g = make_geom(dim, dim);
    const float alpha = (float)(1.0 / sqrt((double)head_size));
    const int att = attention_oproj_form(dim, kv_dim, head_size, n_heads, seq_len_bin, scratch != nullptr, scratch_bytes, split_min, split_chunk);
    if (att < 0) return Q4_ERR_UNSUPPORTED_SIZE;
    const bool split = att >= 2;
    const int nsp = split ? divUp(seq_len_bin, att == 2 ? 128 : 256) : 1;
    const int nw = LA_WAVES;
    AttOprojArgs a = {};
    const int kv_mul = dim / kv_dim;
    a.att = {xb, q, key_cache, value_cache, head_size, kv_mul, kv_dim, pPos, alpha, seq_len_bin, nullptr};
    a.split = {scratch, q, key_cache, value_cache, head_size, kv_mul, kv_dim, pPos, alpha, xb, sync + 128};
    GemvArgs& oa = a.oproj;
    oa.K = dim; oa.N = dim; oa.pw4 = g.pw4; oa.pzh = g.pzh; oa.sh = g.sh; oa.nslots = g.nslots;
    fill_mat(oa.m[0], wo);
    oa.out[0] = x; oa.x = xb; oa.accum = 1; oa.loff = -1;
    a.sync = sync;
    a.nheads = n_heads;
    a.natt = split ? n_heads * nsp : n_heads;
    a.no = dim / (nw * 4);
#ifdef Q4_PROFILING
    a.dbg = g_dbg;
#endif
    const int TS = g.nslots;
    const size_t smem_gemv = (size_t)TS * 256 * 16 + (size_t)TS * 512 + (size_t)TS * 256 * 4 + 16;
    const size_t smem_att = split ? (size_t)(32 + nw * head_size) * 4 : (size_t)(32 + nw * head_size + seq_len_bin) * 4;
    const size_t smem = smem_gemv > smem_att ? smem_gemv : smem_att;
    if (smem > 64 * 1024) return Q4_ERR_UNSUPPORTED_SIZE;
    const dim3 grid(a.natt + a.no), block(nw * 64);
#define Q4_AO(S, H)                                                                            \
    switch (att) {                                                                             \
        case 0: Q4_LAUNCH((attention_oproj_kernel<S, H, 0>), grid, block, smem, a); break;     \
        case 1: Q4_LAUNCH((attention_oproj_kernel<S, H, 1>), grid, block, smem, a); break;     \
        case 2: Q4_LAUNCH((attention_oproj_kernel<S, H, 2>), grid, block, smem, a); break;     \
        default: Q4_LAUNCH((attention_oproj_kernel<S, H, 3>), grid, block, smem, a); break;    \
    }
    if (g.nslots == 2) Q4_AO(2, false) else Q4_AO(3, true)
#undef Q4_AO
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

// words of hand-off state a model needs for this launch: the error + epoch page, then 4 granule vectors of dim/2 x 8 bytes
size_t attention_block_sync_words(int dim, int n_heads) {
    (void)n_heads;
    return 1024 + (size_t)4 * (dim / 2) * 2;
}

// true when this geometry / sequence-length bin has a fused form: multi-head (dim == kv_dim), head 128, the one-block
// attention (bins below the split-context threshold), K = dim in 2 slots or 3 with a shared half slot

bool attention_block_supported(int dim, int kv_dim, int head_size, int seq_len_bin, int split_min) {
    const QGeom g = make_geom(dim, dim);
    const bool slots_ok = g.nslots == 2 || (g.nslots == 3 && g.pw4 - 2 * 64 <= 32);
    return dim == kv_dim && head_size == 128 && seq_len_bin < split_min && seq_len_bin <= 512 && slots_ok && (dim % (LA_WAVES * 4)) == 0 &&
           ((head_size / 2) % (2 * LA_WAVES)) == 0;
}

int launch_attention_block(q4_half* x, q4_half* xb, q4_half* q, q4_half* key_cache, q4_half* value_cache, const q4_half* rms_w,
                           const QWeight* wq, const QWeight* wk, const QWeight* wv, const QWeight* wo, int dim, int n_heads,
                           long long loff, const int* pPos, float rope_theta, const float2* rope_table, int seq_len_bin,
                           unsigned* sync) {
    const int head_size = dim / n_heads;
    const QGeom g = make_geom(dim, dim);
    AttBlockArgs a = {};
    // ---- QKV role (launch_qkv_fused) ----
    GemvArgs& qa = a.qkv;
    qa.K = dim; qa.N = dim; qa.pw4 = g.pw4; qa.pzh = g.pzh; qa.sh = g.sh; qa.nslots = g.nslots;
    fill_mat(qa.m[0], wq); fill_mat(qa.m[1], wk); fill_mat(qa.m[2], wv);
    qa.out[0] = q; qa.out[1] = key_cache; qa.out[2] = value_cache;
    qa.x = x; qa.rms_w = rms_w; qa.pPos = pPos; qa.loff = loff;
    qa.rope = 1; qa.head_size = head_size; qa.rope_theta = rope_theta; qa.rope_table = rope_table;
    qa.early = g_la_early;                         // the first 8-wave block on each CU sends its weight loads early
    // ---- attention role (launch_attention) ----
    a.att = {xb, q, key_cache + loff, value_cache + loff, head_size, 1, dim, pPos,
             (float)(1.0 / sqrt((double)head_size)), seq_len_bin, nullptr};                 // llama2_q4.cu:273
    // ---- o-proj role (q4_matmul_q4(x, xb, wo, dim, dim, accum)) ----
    GemvArgs& oa = a.oproj;
    oa.K = dim; oa.N = dim; oa.pw4 = g.pw4; oa.pzh = g.pzh; oa.sh = g.sh; oa.nslots = g.nslots;
    fill_mat(oa.m[0], wo);
    oa.out[0] = x; oa.x = xb; oa.accum = 1; oa.loff = -1;
    a.sync = sync;
#ifdef Q4_PROFILING
    a.dbg = g_dbg;
#endif
    a.nqx = dim / (LA_WAVES * 4);
    a.nq = 3 * a.nqx;
    a.nheads = n_heads;
    a.no = dim / (LA_WAVES * 4);
    const int TS = g.nslots;
    const size_t smem_gemv = (size_t)TS * 256 * 16 + (size_t)TS * 512 + (size_t)TS * 256 * 4 + 16 + 4 * LA_WAVES * 2;   // + the producers' 2 x 16 halves
    const size_t smem_att = (size_t)(32 + LA_WAVES * head_size + seq_len_bin + 3 * 64) * 4;   // + q / k row / v row of the head
    const size_t smem = smem_gemv > smem_att ? smem_gemv : smem_att;
    const dim3 grid(a.nq + a.nheads + a.no), block(LA_WAVES * 64);
    if (g.nslots == 2) {
        if (seq_len_bin <= 128) Q4_LAUNCH((attention_block_kernel<2, false, 4>), grid, block, smem, a);
        else Q4_LAUNCH((attention_block_kernel<2, false, 8>), grid, block, smem, a);
    } else {
        if (seq_len_bin <= 128) Q4_LAUNCH((attention_block_kernel<3, true, 4>), grid, block, smem, a);
        else Q4_LAUNCH((attention_block_kernel<3, true, 8>), grid, block, smem, a);
    }
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // namespace q4
