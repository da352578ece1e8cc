// layer_attn_h256.hip -- head-256 instantiations of the attention -> o-proj launch (layer_attn.h)
#include "layer_attn.h"
namespace q4 {
int launch_attention_oproj_h256(int slots_kind, int att, dim3 grid, dim3 block, size_t smem, const AttOprojArgs& a, int* max_blocks_per_cu)
    Q4_AO_DISPATCH(32)
}  // namespace q4
