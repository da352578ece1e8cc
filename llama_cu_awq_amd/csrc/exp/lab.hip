// exp/lab.hip -- the LABORATORY translation unit: linked into libllama2_q4_prof.so only (csrc/Makefile), never into libllama2_q4.so. It holds every
// experiment form that was built, measured and NOT shipped (EXPERIMENTS.md has the numbers) and registers them in the hook table g_lab
// (q4_internal.h), which the product's dispatchers consult first; tests/prof_cases.py holds each of them to the shipped kernels' bits, tools/ time them.
//   exp/ffn_engine.h          gate/up as a loader / consumer engine on LDS-DMA              knob 11 = 1..6
//   exp/ffn_strip_variants.h  gate/up strips: ring depth 4 / 8, paced issue, stamped build  knob 11 = 8..14 (and the stamped product form when a debug buffer is set)
//   exp/qkv_strip.h           the 13B q/k/v launch as strips                                knob 11 = 8
//   exp/cls_argmax.h          the greedy sampler as the classifier launch's epilogue        knob 12
//   exp/attention_ring.h      split-context attention: K / V rows on LDS-DMA rings          knob 14 (included by layer_attn.h in this build)
//   gate/up ablations of the wave-owned kernel (gemv_q4.h, ABL 1..4)                        q4_set_ablate
#include "ffn_engine.h"
#include "ffn_strip_variants.h"
#include "qkv_strip.h"
#include "cls_argmax.h"

namespace q4 {

int g_cls_argmax = 0;    // knob 12

static bool strips_variant_form() {
    return g_gemv_form >= GEMV_STRIPS_EVERYWHERE && g_gemv_form <= GEMV_STRIPS_PACED_D8_ROTATED && g_gemv_form != 11;
}
static bool lab_ffn_covers(const GemvArgs& a) {
    if (g_ablate) return pick_slots(a.nslots) == 2 && a.rms_w != nullptr && g_ablate >= 1 && g_ablate <= 4;   // ablations of the 7B gate/up kernel
    if (ffn_engine_covers(a)) return true;
    if (strips_variant_form()) return ffn_strip_shape(a);                       // a strips variant wherever the shape is covered
    return a.dbg != nullptr && !strip_k5120(a) && ffn_strip_covers(a);          // tools/lab/timeline_strip.py: the product's form, stamped
}
template <bool NORM, bool STAMPS>
static int launch_strip_setting(const GemvArgs& a) {
    switch (g_gemv_form) {
        case GEMV_STRIPS_D4: return launch_strip_variant<NORM, 4, 0, STAMPS>(a);
        case GEMV_STRIPS_D8: return launch_strip_variant<NORM, 8, 0, STAMPS>(a);
        case GEMV_STRIPS_PACED_D4: return launch_strip_variant<NORM, 4, 1, STAMPS>(a);
        case GEMV_STRIPS_PACED_D8: return launch_strip_variant<NORM, 8, 1, STAMPS>(a);
        case GEMV_STRIPS_PACED_D8_ROTATED: return launch_strip_variant<NORM, 8, 2, STAMPS>(a);
        default: return launch_strip_variant<NORM, 2, 0, STAMPS>(a);          // the product's form (GEMV_PRODUCT with stamps, GEMV_STRIPS_EVERYWHERE)
    }
}
static int lab_ffn_launch(const GemvArgs& a, int waves) {
    if (g_ablate == 1) return launch_one<MODE_FFN, 2, 2, true, 1>(a, waves);
    if (g_ablate == 2) return launch_one<MODE_FFN, 2, 2, true, 2>(a, waves);
    if (g_ablate == 3) return launch_one<MODE_FFN, 2, 2, true, 3>(a, waves);
    if (g_ablate == 4) return launch_one<MODE_FFN, 2, 2, true, 4>(a, waves);
    if (ffn_engine_covers(a)) return launch_ffn_engine(a);
    const bool norm = a.rms_w != nullptr;
    if (g_gemv_form != GEMV_STRIPS_D4 && strip_pairs(a)) return launch_strip_pair(a);      // the product's pair-unit kernel
    if (strip_k5120(a)) {
        if (g_gemv_form == GEMV_STRIPS_D4) return norm ? launch_strip_variant<true, 4, 0, false, 3>(a) : launch_strip_variant<false, 4, 0, false, 3>(a);
        return norm ? launch_strip<true, 3>(a) : launch_strip<false, 3>(a);
    }
    if (a.dbg) return norm ? launch_strip_setting<true, true>(a) : launch_strip_setting<false, true>(a);
    return norm ? launch_strip_setting<true, false>(a) : launch_strip_setting<false, false>(a);
}

static bool lab_qkv_covers(const GemvArgs& a) { return qkv_strip_covers(a); }
static int lab_qkv_launch(const GemvArgs& a) { return launch_qkv_strip(a); }

template <int NS>
static int launch_cls_strip_argmax(q4_half* out, const q4_half* x, const q4_half* rms_w, const q4_half* w, int n, int d, const ClsArgmax& am) {
    constexpr size_t smem = StripClsLds<NS, CLS_D>::BYTES;
    { const int rc = lds_opt_in((const void*)cls_strip_argmax_kernel<NS, CLS_D>, smem); if (rc) return rc; }     // (knob 12 is set before the graphs are captured)
    const unsigned nb = (unsigned)cu_count();
    Q4_LAUNCH((cls_strip_argmax_kernel<NS, CLS_D>), dim3(nb), dim3(STRIP_WAVES * 64), smem, reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(rms_w),
              (const void*)w, (unsigned)((size_t)d * n * 2), (unsigned)d / nb, (unsigned)d % nb, out, n, (unsigned)n * 2u, am);
    Q4_LAUNCH_CHECK();
    return Q4_OK;
}
static bool lab_cls_argmax(q4_half* logits, const q4_half* x, const q4_half* rms_w, const q4_half* wcls, int dim, int vocab, const GreedyTail* tail, int* rc) {
    if (!g_cls_argmax || rms_w == nullptr || tail->words == nullptr || (unsigned)cu_count() > CLS_SYNC_BLOCKS) return false;
    const ClsArgmax am = {tail->words, reinterpret_cast<unsigned long long*>(tail->words + 2), tail->result, tail->pPos, tail->pPosGpu, tail->write_token,
                          tail->x_next, tail->table, dim};
    *rc = dim == 4096 ? launch_cls_strip_argmax<8>(logits, x, rms_w, wcls, dim, vocab, am) : launch_cls_strip_argmax<10>(logits, x, rms_w, wcls, dim, vocab, am);
    return true;
}

static const int g_lab_registered = (g_lab = LabHooks{lab_ffn_covers, lab_ffn_launch, lab_qkv_covers, lab_qkv_launch, lab_cls_argmax}, 0);

}  // namespace q4
