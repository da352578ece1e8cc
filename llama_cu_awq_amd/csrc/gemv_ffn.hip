// gemv_ffn.hip -- instantiations of the int4 GEMV for ffn_matvec_silu_kernel (gpu_kernels.h:256-275)
#include "gemv_q4.h"
namespace q4 {
bool ffn_strips_cover(const GemvArgs& a);   // gemv_ffn_strip.hip: the strips form (gemv_strip.h), for wide matrices
int launch_ffn_strips(const GemvArgs& a);
int launch_gemv_ffn(const GemvArgs& a, int cols, int waves) {
    if (g_lab.ffn_covers && g_lab.ffn_covers(a)) return g_lab.ffn_launch(a, waves);   // (null in the shipped library: q4_internal.h)
    if (ffn_strips_cover(a)) return launch_ffn_strips(a);
#define Q4_CASE(S, C) if (slots == S && cols == C) { \
        return a.rms_w ? launch_one<MODE_FFN, S, C, true>(a, waves) : launch_one<MODE_FFN, S, C, false>(a, waves); }
    const int slots = pick_slots(a.nslots);
    if (slots == 3 && a.nslots == 3 && cols == 2 && half_tail(a))
        return a.rms_w ? launch_one<MODE_FFN, 3, 2, true, 0, 1, true>(a, waves) : launch_one<MODE_FFN, 3, 2, false, 0, 1, true>(a, waves);
    if (cols != 2 && cols != 4) cols = 2;
    if (slots >= 4 && cols == 4) cols = 2;   // gate+up doubles the loads in flight
    Q4_CASE(2, 2) Q4_CASE(2, 4)
    Q4_CASE(3, 2) Q4_CASE(3, 4)
    Q4_CASE(4, 2) Q4_CASE(6, 2) Q4_CASE(7, 2) Q4_CASE(8, 2)
#undef Q4_CASE
    return Q4_ERR_UNSUPPORTED_SIZE;
}
}  // namespace q4
