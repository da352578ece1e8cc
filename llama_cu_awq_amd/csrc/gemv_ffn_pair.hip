// gemv_ffn_pair.hip -- host side of the FFN half of a layer as one launch (gemv_ffn_pair.h; fusion level 4, llama2_q4.cu:326-332).
#include "gemv_ffn_pair.h"

namespace q4 {

int g_fp_pre = 3;      // profiling knob 16: gather mode (gemv_ffn_pair.h, FfnPairArgs::pre)
int g_fp_mute = 0;
int g_fp_nt = 0;       // profiling knob 18: the gather's loads nt (past the L1, served by the XCD's L2) instead of sc1     // profiling knob 17: the blocks of the next n launches do not publish (a real time-out)

static void fill_mat(GemvMat& m, const QWeight* w) { m.w = w->weight; m.z = w->zeros; m.s = w->scales; }
static void fill_gate_up(GemvArgs& a, int dim, int hidden) {
    const QGeom g = make_geom(dim, hidden);
    a.K = dim; a.N = hidden; a.pw4 = g.pw4; a.pzh = g.pzh; a.sh = g.sh; a.nslots = g.nslots;
}

// words the launch needs behind the model's other hand-off words: hidden / 2 granules of 8 bytes
size_t ffn_pair_sync_words(int hidden) { return (size_t)(hidden / 2) * 2; }

bool ffn_pair_covers(int dim, int hidden) {
    if ((dim & 7) || (hidden & 31)) return false;
    GemvArgs a = {};
    fill_gate_up(a, dim, hidden);
    return ffn_pair_shape(a, hidden, dim);
}

// 146 KiB of LDS: the opt-in is not a stream operation -- build_transformer makes it for the model, outside any capture
int ffn_pair_prepare() {
    int rc = lds_opt_in((const void*)ffn_pair_kernel<true, false>, FfnPairLds::BYTES);
#ifdef Q4_PROFILING
    if (!rc) rc = lds_opt_in((const void*)ffn_pair_kernel<true, true>, FfnPairLds::BYTES);

#endif
    return rc;
}

int launch_ffn_pair(q4_half* x, q4_half* hb, const q4_half* rms_w, const QWeight* gate, const QWeight* up, const QWeight* down, int dim, int hidden,
                    unsigned* sync, size_t gran_word, unsigned tag_add) {
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
    const unsigned pairs = (unsigned)hidden / 2u;
#ifdef Q4_PROFILING
    if (g_fp_mute > 0) { p.mute = 1; g_fp_mute--; }
    p.dbg = g_dbg;
#define FP_GO(...) do { Q4_LAUNCH((ffn_pair_kernel<__VA_ARGS__>), dim3(nb), dim3(STRIP_WAVES * 64), FfnPairLds::BYTES, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w), \
                  (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), pairs / nb, pairs % nb, a, p); Q4_LAUNCH_CHECK(); return Q4_OK; } while (0)
    if (g_dbg) {
        Q4_LAUNCH((ffn_pair_kernel<true, true>), dim3(nb), dim3(STRIP_WAVES * 64), FfnPairLds::BYTES, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
                  (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), pairs / nb, pairs % nb, a, p);
        Q4_LAUNCH_CHECK();
        return Q4_OK;
    }
#endif
    Q4_LAUNCH((ffn_pair_kernel<true, false>), dim3(nb), dim3(STRIP_WAVES * 64), FfnPairLds::BYTES, reinterpret_cast<const u32x4*>(a.x), reinterpret_cast<const u32x4*>(a.rms_w),
              (const void*)a.m[0].w, (const void*)a.m[1].w, (unsigned)(a.N * a.pw4 * 16), pairs / nb, pairs % nb, a, p);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}

}  // namespace q4
