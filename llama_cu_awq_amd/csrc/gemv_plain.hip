// gemv_plain.hip -- instantiations of the int4 GEMV for mat_vec_kernel_int4 (gpu_kernels.h:235-240)
#include "gemv_strip_down.h"   // (gemv_q4.h, the LDS-DMA helpers) + the 13B down projection as strips
namespace q4 {
int launch_gemv_plain(const GemvArgs& a, int cols, int waves) {
#define Q4_CASE(S, C) if (slots == S && cols == C) return launch_one<MODE_PLAIN, S, C, false>(a, waves);
    const int slots = pick_slots(a.nslots);
    if ((g_ksplit && a.nslots >= 5) || a.nslots > 8) {   // long-K (down projection): two waves per column group (gemv_q4.h, KS)
        // (three balanced parts were measured too, in 3 / 6 / 12-wave blocks: 7B 7.5-8.4 us against 6.6, 13B 13.1-15.7 against 12.0)
        const int ks = 2;
        GemvArgs b = a;
        b.ku = (divUp(a.pw4, ks) + 3) & ~3;                    // whole quantisation groups per k-part
        const int sh = divUp(b.ku, 64);                        // up to 8 slots (K <= 32768, Llama-2-70B / CodeLlama-34B down projections)
        // a k-part whose last slot holds at most 32 units shares it between the columns of a pair (13B: 216 = 3 x 64 + 24)
        if (g_half_tail && g_ksplit != 4 && b.ku - (sh - 1) * 64 <= 32) {
            if (down_strip_covers(b, true)) return launch_down_strip<4, true>(b);   // the same arithmetic, every CU the same bytes (13B: 9.83 -> 8.64 us per launch)
#define Q4_KSH(S) if (sh == S) return launch_one<MODE_PLAIN, S, 4, false, 0, 2, true>(b, waves);
            Q4_KSH(2) Q4_KSH(3) Q4_KSH(4) Q4_KSH(5) Q4_KSH(6) Q4_KSH(7) Q4_KSH(8)
#undef Q4_KSH
        }
        if (g_ksplit != 4 && down_strip_covers(b, false)) return launch_down_strip<3, false>(b);
#define Q4_KS(S) if (sh == S) return launch_one<MODE_PLAIN, S, 4, false, 0, 2>(b, waves);
        Q4_KS(1) Q4_KS(2) Q4_KS(3) Q4_KS(4) Q4_KS(5) Q4_KS(6) Q4_KS(7) Q4_KS(8)
#undef Q4_KS
        return Q4_ERR_UNSUPPORTED_SIZE;
    }
    if (cols == 4 && slots == 3 && a.nslots == 3 && half_tail(a)) {   // K = 5120 (13B o-proj and unfused q/k/v): shared half slot
        if (divUp(a.N, 4 * waves) <= 320) return launch_one<MODE_PLAIN, 3, 4, false, 5, 1, true>(a, waves);
        return launch_one<MODE_PLAIN, 3, 4, false, 0, 1, true>(a, waves);
    }
    if (cols == 4 && slots <= 3 && divUp(a.N, 4 * waves) <= 320) {   // about one block per CU: every load first
        if (slots == 2) return launch_one<MODE_PLAIN, 2, 4, false, 5>(a, waves);
        if (slots == 3) return launch_one<MODE_PLAIN, 3, 4, false, 5>(a, waves);
    }
    if (cols != 4 && cols != 8) cols = 4;
    if (slots >= 4 && cols == 8) cols = 4;   // 8 columns x >= 4 uint4 per lane does not fit the register file
    Q4_CASE(2, 4) Q4_CASE(2, 8)
    Q4_CASE(3, 4) Q4_CASE(3, 8)
    Q4_CASE(4, 4) Q4_CASE(6, 4) Q4_CASE(7, 4) Q4_CASE(8, 4)
#undef Q4_CASE
    return Q4_ERR_UNSUPPORTED_SIZE;
}
}  // namespace q4
