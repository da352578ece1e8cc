// gemv_ffn_strip.hip -- the gate/up launch as strips (gemv_strip.h: sixteen self-loading waves per CU on LDS-DMA rings), the form that ships for
// wide matrices; launch_gemv_ffn (gemv_ffn.hip) asks here first. Also home of the process's GEMV-form word and of the laboratory hook table, both
// inert in the shipped library (q4_internal.h).
#include "gemv_strip.h"

namespace q4 {

int g_gemv_form = GEMV_PRODUCT;
LabHooks g_lab = {};

bool ffn_strips_cover(const GemvArgs& a) { return ffn_strip_covers(a); }
int launch_ffn_strips(const GemvArgs& a) { return launch_ffn_strip(a); }

}  // namespace q4
