// gemv_qkv.hip -- instantiations of the int4 GEMV for qkv_matvec_kernel (gpu_kernels.h:242-254)
#include "gemv_q4.h"
namespace q4 {
int launch_gemv_qkv(const GemvArgs& a, int cols, int waves) {
#define Q4_CASE(S, C) if (slots == S && cols == C) { \
        return a.rms_w ? launch_one<MODE_QKV, S, C, true>(a, waves) : launch_one<MODE_QKV, S, C, false>(a, waves); }
    const int slots = pick_slots(a.nslots);
    if (cols != 2 && cols != 4) cols = 2;
    if (slots >= 6 && cols == 4) cols = 2;
    Q4_CASE(2, 2) Q4_CASE(2, 4)
    Q4_CASE(3, 2) Q4_CASE(3, 4)
    Q4_CASE(4, 2) Q4_CASE(4, 4)
    Q4_CASE(6, 2) Q4_CASE(7, 2) Q4_CASE(8, 2)
#undef Q4_CASE
    return Q4_ERR_UNSUPPORTED_SIZE;
}
}  // namespace q4
