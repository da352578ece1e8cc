// gemv_ffn_pair.hip -- host side of the FFN half of a layer as one launch (gemv_ffn_pair.h; fusion level 4, llama2_q4.cu:326-332).
#include "gemv_ffn_pair.h"
#include <math.h>

namespace q4 {

int g_fp_pre = 3;      // profiling knob 16: gather mode (gemv_ffn_pair.h, FfnPairArgs::pre)
int g_fp_mute = 0;
int g_fp_nt = 0;       // profiling knob 18: the gather's loads nt (past the L1, served by the XCD's L2) instead of sc1     // profiling knob 17: the blocks of the next n launches do not publish (a real time-out)

static void fill_mat(GemvMat& m, const QWeight* w) { m.w = w->weight; m.z = w->zeros; m.s = w->scales; }
static void fill_gate_up(GemvArgs& a, int dim, int hidden) {
    const QGeom g = make_geom(dim, hidden);
    a.K = dim; a.N = hidden; a.pw4 = g.pw4; a.pzh = g.pzh; a.sh = g.sh; a.nslots = g.nslots;
}

// words the launch needs behind the model's other hand-off words: hidden / 2 granules of 8 bytes for hb, then dim / 2 for the residual stream
// between its second and third phase (fusion level 5)
// between its second and third phase (fusion level 5), then dim / 2 for the residual stream between the output projection and the FFN half (level 6)
size_t ffn_pair_sync_words(int dim, int hidden) { return (size_t)(hidden / 2) * 2 + 2 * gran_area_words((size_t)dim / 2); }   // hb dense; the two x vectors spread (gemv_q4.h, gran_slot)

// fusion level 6: the launch begins with the layer's attention and output projection. Llama-2-7B's shape below the split-context bins: 128-wide heads (four
// 64-byte V slices each), as many (head, slice) units as half the blocks -- the other half are the output projection, 32 columns of two k-slots each; the attention's arithmetic is
// the V-slice role's of layer_attn.h (forms 5 / 6), which q4_runtime.hip checks is what fusion level 3 would run
bool layer_att_covers(int dim, int hidden, int kv_dim, int n_heads, int seq_len_bin) {
    if (!ffn_pair_covers(dim, hidden) || n_heads < 1 || dim % n_heads || kv_dim != dim) return false;
    const int nb = cu_count(), head_size = dim / n_heads;
    // (every wave then has ten gate/up pieces or more; the pieces requested ahead fit around the down columns 8 .. 15)
    if (!FfnPairLds::credit_fits((unsigned)make_geom(hidden, dim).pw4 * 16u)) return false;
    return dim == 4096 && head_size == 128 && n_heads * 8 == nb && dim == 16 * nb && seq_len_bin <= 256 && g_ao_vslice != 0 && (hidden / 2) / nb >= 20;
}

// fusion level 5: the launch also runs rmsnorm + q/k/v + RoPE + KV write of the NEXT layer. Multi-head models with 128-wide heads and a rotation table
// (every Llama-2-7B-shaped model): a block's eight RoPE pairs stay inside one head.
bool ffn_qkv_covers(int dim, int hidden, int kv_dim, int head_size, bool have_rope_table) {
    if (!ffn_pair_covers(dim, hidden) || kv_dim != dim || !have_rope_table || head_size < 16 || (head_size & 1)) return false;
    const int nb = cu_count(), hp = head_size / 2;
    if ((dim / 2) % nb) return false;
    const int ppb = dim / 2 / nb;
    return ppb == 8 && hp % ppb == 0 && dim / nb == 16 && 48 * 2048 <= (int)FfnPairLds::DW_BYTES;
}

bool ffn_pair_covers(int dim, int hidden) {
    if ((dim & 7) || (hidden & 31)) return false;
    GemvArgs a = {};
    fill_gate_up(a, dim, hidden);
    return ffn_pair_shape(a, hidden, dim);
}

// 146 KiB of LDS: the opt-in is not a stream operation -- build_transformer makes it for the model, outside any capture
int ffn_pair_prepare() {
    static int prepared_device = -1;       // (called in front of every launch: one hipGetDevice and a comparison once the device's opt-ins are made)
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev == prepared_device) return Q4_OK;
    int rc = lds_opt_in((const void*)ffn_pair_kernel<true, false>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, false, true>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, false, true, 1>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, false, true, 2>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, false, false, 1>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, false, false, 2>, FfnPairLds::BYTES);
#ifdef Q4_PROFILING
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, true>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, true, true>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, true, true, 1>, FfnPairLds::BYTES);
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, true, true, 2>, FfnPairLds::BYTES);
#endif
    if (!rc) prepared_device = dev;
    return rc;
}

int launch_ffn_pair(q4_half* x, q4_half* hb, const q4_half* rms_w, const QWeight* gate, const QWeight* up, const QWeight* down, int dim, int hidden,
                    unsigned* sync, size_t gran_word, unsigned tag_add, const FfnQkvNext* next, const FfnLayerAtt* att) {
    if (!rms_w || !sync || !ffn_pair_covers(dim, hidden)) return Q4_ERR_UNSUPPORTED_SIZE;
    { const int rc = ffn_pair_prepare(); if (rc) return rc; }
    GemvArgs a = {};
    fill_gate_up(a, dim, hidden);
    fill_mat(a.m[0], gate); fill_mat(a.m[1], up);
    a.out[0] = hb; a.x = x; a.rms_w = rms_w;
    const QGeom g = make_geom(hidden, dim);
    const unsigned nb = (unsigned)cu_count();
    FfnPairArgs p = {};
    fill_mat(p.d, down);
    p.xio = x;
    p.sync = sync;
    p.error = sync + SYNC_ERROR;
    p.gran = reinterpret_cast<u32x2v*>(sync + gran_word);
    p.Kd = hidden; p.Nd = dim; p.pw4 = g.pw4; p.pzh = g.pzh; p.sh = g.sh;
    p.ku = (divUp(g.pw4, 2) + 3) & ~3;                       // whole quantisation groups per k-part (gemv_plain.hip)
    p.dbase = (unsigned)dim / nb; p.drem = (unsigned)dim % nb;
    p.pre = (unsigned)g_fp_pre;
    p.tag_add = tag_add;
    QkvNextArgs q3 = {};
    if (next) {
        const int kv_dim = next->kv_dim, head_size = next->head_size;
        if (!ffn_qkv_covers(dim, hidden, kv_dim, head_size, next->rope_table != nullptr)) return Q4_ERR_UNSUPPORTED_SIZE;
        fill_mat(q3.m[0], next->wq); fill_mat(q3.m[1], next->wk); fill_mat(q3.m[2], next->wv);
        q3.rms_w = next->rms_w; q3.q = next->q; q3.kc = next->kc; q3.vc = next->vc; q3.pPos = next->pPos; q3.rope_table = next->rope_table;
        q3.xgran = reinterpret_cast<u32x2v*>(sync + gran_word + (size_t)(hidden / 2) * 2);
        q3.bump = next->bump; q3.head_size = head_size; q3.kv_dim = kv_dim; q3.ppb = (unsigned)(dim / 2) / nb;
    }
    LayerAttArgs la = {};
    if (att) {
        if (tag_add != 0u || !layer_att_covers(dim, hidden, att->kv_dim, att->n_heads, att->seq_len_bin)) return Q4_ERR_UNSUPPORTED_SIZE;
        const int head_size = dim / att->n_heads;
        la.att = {att->xb, att->q, att->kc, att->vc, head_size, dim / att->kv_dim, att->kv_dim, att->pPos, (float)(1.0 / sqrt((double)head_size)), att->seq_len_bin, nullptr};
        fill_mat(la.o, att->wo);
        la.agran = reinterpret_cast<u32x2v*>(sync + SYNC_GRANULES);
        la.xogran = reinterpret_cast<u32x2v*>(sync + gran_word + (size_t)(hidden / 2) * 2 + gran_area_words((size_t)dim / 2));
        la.nheads = (unsigned)att->n_heads; la.natt = 4u * (unsigned)att->n_heads;
    }
    const unsigned pairs = (unsigned)hidden / 2u;
#define FP_GO(...) do { Q4_LAUNCH((ffn_pair_kernel<__VA_ARGS__>), dim3(nb), dim3(STRIP_WAVES * 64), FfnPairLds::BYTES, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w), \
                  (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), pairs / nb, pairs % nb, a, p, q3, la); Q4_LAUNCH_CHECK(); return Q4_OK; } while (0)
#ifdef Q4_PROFILING
    if (g_fp_mute > 0) { p.mute = 1; g_fp_mute--; }
    p.dbg = g_dbg;
    if (g_dbg && next && att) { if (att->seq_len_bin <= 128) FP_GO(true, true, true, 1); else FP_GO(true, true, true, 2); }
    if (g_dbg && next && g_fusion < 6) FP_GO(true, true, true);
    if (g_dbg && g_fusion < 5) FP_GO(true, true, false);        // (at level 5 the stamps are the three-phase launches': the last layer's plain pair launch does not overwrite them)
#endif
    if (att) {
        const bool b128 = att->seq_len_bin <= 128;
        if (next) { if (b128) FP_GO(true, false, true, 1); else FP_GO(true, false, true, 2); }
        if (b128) FP_GO(true, false, false, 1); else FP_GO(true, false, false, 2);
    }
    if (next) FP_GO(true, false, true);
    FP_GO(true, false, false);
#undef FP_GO
}

}  // namespace q4
