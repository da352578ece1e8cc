// gemv_ffn.hip -- instantiations of the int4 GEMV for ffn_matvec_silu_kernel (gpu_kernels.h:256-275)
#include "gemv_q4.h"
namespace q4 {
int launch_gemv_ffn(const GemvArgs& a, int cols, int waves) {
#define Q4_CASE(S, C) if (slots == S && cols == C) { \
        return a.rms_w ? launch_one<MODE_FFN, S, C, true>(a, waves) : launch_one<MODE_FFN, S, C, false>(a, waves); }
    const int slots = pick_slots(a.nslots);
    if (cols != 1 && cols != 2 && cols != 4) cols = 2;
    if (slots >= 3 && cols == 4) cols = 2;   // gate+up doubles the loads in flight
    if (slots >= 6 && cols == 2) cols = 1;
    Q4_CASE(2, 1) Q4_CASE(2, 2) Q4_CASE(2, 4)
    Q4_CASE(3, 1) Q4_CASE(3, 2)
    Q4_CASE(4, 1) Q4_CASE(4, 2)
    Q4_CASE(6, 1) Q4_CASE(7, 1) Q4_CASE(8, 1)
#undef Q4_CASE
    return Q4_ERR_UNSUPPORTED_SIZE;
}
}  // namespace q4
