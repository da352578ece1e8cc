// exp/lab.hip -- the LABORATORY translation unit: linked into libllama2_q4_prof.so only (csrc/Makefile), never into libllama2_q4.so. It holds every
// experiment form that was built, measured and NOT shipped (EXPERIMENTS.md has the numbers) and registers them in the hook table g_lab
// (q4_internal.h), which the product's dispatchers consult first; tests/prof_cases.py holds each of them to the shipped kernels' bits, tools/ time them.
//   exp/ffn_strip_variants.h  gate/up strips: ring depth 4 / 8, paced issue, stamped build  knob 11 = 8..14 (and the stamped product form when a debug buffer is set)
//   gate/up ablations of the wave-owned kernel (gemv_q4.h, ABL 1..4)                        q4_set_ablate
// Round 6 removed the forms whose negative is conclusive (loader / consumer engine #12, q/k/v strips #20, the sampler-epilogue classifier #19, K / V rings
// #29): EXPERIMENTS.md keeps their records, git history their code (commit 3718b50 is the last that holds them).
#include "ffn_strip_variants.h"

namespace q4 {

static bool strips_variant_form() {
    return g_gemv_form >= GEMV_STRIPS_EVERYWHERE && g_gemv_form <= GEMV_STRIPS_PACED_D8_ROTATED && g_gemv_form != 11;
}
static bool lab_ffn_covers(const GemvArgs& a) {
    if (g_ablate) return pick_slots(a.nslots) == 2 && a.rms_w != nullptr && g_ablate >= 1 && g_ablate <= 4;   // ablations of the 7B gate/up kernel
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
    const bool norm = a.rms_w != nullptr;
    if (g_gemv_form != GEMV_STRIPS_D4 && strip_pairs(a)) return launch_strip_pair(a);      // the product's pair-unit kernel
    if (strip_k5120(a)) {
        if (g_gemv_form == GEMV_STRIPS_D4) return norm ? launch_strip_variant<true, 4, 0, false, 3>(a) : launch_strip_variant<false, 4, 0, false, 3>(a);
        return norm ? launch_strip<true, 3>(a) : launch_strip<false, 3>(a);
    }
    if (a.dbg) return norm ? launch_strip_setting<true, true>(a) : launch_strip_setting<false, true>(a);
    return norm ? launch_strip_setting<true, false>(a) : launch_strip_setting<false, false>(a);
}

static const int g_lab_registered = (g_lab = LabHooks{lab_ffn_covers, lab_ffn_launch}, 0);

}  // namespace q4
