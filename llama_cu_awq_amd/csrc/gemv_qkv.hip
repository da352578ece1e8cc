// gemv_qkv.hip -- instantiations of the int4 GEMV for qkv_matvec_kernel (gpu_kernels.h:242-254)
#include "gemv_q4.h"
namespace q4 {
int launch_gemv_qkv(const GemvArgs& a, int cols, int waves) {
#define Q4_CASE(S) if (slots == S) { \
        return a.rms_w ? launch_one<MODE_QKV, S, 4, true>(a, waves) : launch_one<MODE_QKV, S, 4, false>(a, waves); }
    const int slots = pick_slots(a.nslots);
    (void)cols;   // 4 columns (two RoPE pairs) per wave
    if (slots == 3 && a.nslots == 3 && half_tail(a))
        return a.rms_w ? launch_one<MODE_QKV, 3, 4, true, 0, 1, true>(a, waves) : launch_one<MODE_QKV, 3, 4, false, 0, 1, true>(a, waves);
    Q4_CASE(2) Q4_CASE(3) Q4_CASE(4) Q4_CASE(6) Q4_CASE(7) Q4_CASE(8)
#undef Q4_CASE
    return Q4_ERR_UNSUPPORTED_SIZE;
}
}  // namespace q4
