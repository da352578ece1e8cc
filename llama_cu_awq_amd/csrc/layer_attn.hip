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

struct AttBlockArgs {
    GemvArgs qkv;
    GemvArgs oproj;
    AttArgs att;
    unsigned* sync;                  // [heads] q/k/v arrivals per head, [heads] attention arrivals, [heads+1] finished
                                     // o-proj blocks, [heads+2] error flag
    unsigned nq, nqx, nheads, no;    // QKV blocks (nqx per matrix), attention blocks, o-proj blocks
    unsigned qkv_target;             // producer blocks per head: 3 * (head_size/2) / (2 * LA_WAVES)
};

template <int SLOTS, bool HALF, int U>
__global__ void __launch_bounds__(LA_WAVES * 64) attention_block_kernel(const AttBlockArgs a) {
    const unsigned b = blockIdx.x;
    Handoff ho = {};
    ho.error = a.sync + a.nheads + 2;
    if (b < a.nq) {
        ho.signal = a.sync;
        gemv_q4_body<MODE_QKV, SLOTS, 4, true, 0, 1, HALF, ROLE_PRODUCER>(a.qkv, b % a.nqx, b / a.nqx, ho);
    } else if (b < a.nq + a.nheads) {
        ho.wait = a.sync;
        ho.wait_target = a.qkv_target;
        ho.signal = a.sync + a.nheads;
        attention_body<16, U, LA_WAVES, true>(a.att, (int)(b - a.nq), ho);
    } else {
        ho.wait = a.sync + a.nheads;
        ho.wait_target = a.nheads;
        ho.done = a.sync + a.nheads + 1;
        ho.done_target = a.no;
        ho.clear = a.sync;
        ho.clear_n = (int)a.nheads + 2;
        gemv_q4_body<MODE_PLAIN, SLOTS, 4, false, 5, 1, HALF, ROLE_CONSUMER>(a.oproj, b - a.nq - a.nheads, 0, ho);
    }
}

static void fill_mat(GemvMat& m, const QWeight* w) { m.w = w->weight; m.z = w->zeros; m.s = w->scales; }

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
    qa.early = 8;                                  // the first 8-wave block on each CU sends its weight loads early
    // ---- attention role (launch_attention) ----
    a.att = {xb, q, key_cache + loff, value_cache + loff, head_size, 1, dim, pPos,
             (float)(1.0 / sqrt((double)head_size)), seq_len_bin, nullptr};                 // llama2_q4.cu:273
    // ---- o-proj role (q4_matmul_q4(x, xb, wo, dim, dim, accum)) ----
    GemvArgs& oa = a.oproj;
    oa.K = dim; oa.N = dim; oa.pw4 = g.pw4; oa.pzh = g.pzh; oa.sh = g.sh; oa.nslots = g.nslots;
    fill_mat(oa.m[0], wo);
    oa.out[0] = x; oa.x = xb; oa.accum = 1; oa.loff = -1;
    a.sync = sync;
    a.nqx = dim / (LA_WAVES * 4);
    a.nq = 3 * a.nqx;
    a.nheads = n_heads;
    a.no = dim / (LA_WAVES * 4);
    a.qkv_target = 3 * (head_size / 2) / (2 * LA_WAVES);
    const int TS = g.nslots;
    const size_t smem_gemv = (size_t)TS * 256 * 16 + (size_t)TS * 512 + (size_t)TS * 256 * 4 + 16;
    const size_t smem_att = (size_t)(32 + LA_WAVES * head_size + seq_len_bin) * 4;
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
