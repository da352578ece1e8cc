// llama2_q4.hpp -- the reference's host functions under their own names (llama2_q4.cu:209-284, 286-395, 408-432),
// as thin C++ wrappers over the C ABI. Error behaviour of the reference restored: print its message, exit(EXIT_FAILURE).
// `half` is whatever 16-bit float type the including TU uses (hip_fp16.h's __half, _Float16, uint16_t): only pointers cross.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include "llama2_q4.h"

#ifndef Q4_HALF_T
#define Q4_HALF_T q4_half
#endif

namespace llama2_q4 {
typedef Q4_HALF_T half_t;

inline void die_on(int rc) {
    if (rc) { printf("\n%s\n", q4_status_string(rc)); exit(EXIT_FAILURE); }
}
inline void rmsnorm(half_t* o, half_t* x, half_t* weight, int size) {                                    // :209
    die_on(q4_rmsnorm((q4_half*)o, (const q4_half*)x, (const q4_half*)weight, size));
}
inline void matmul(half_t* xout, half_t* x, half_t* w, int n, int d, int batch = 1, int x_stride = 0, int w_stride = 0,
                   int op_stride = 0, int w_row_stride = -1, float alpha = 1.0f) {                         // :214
    die_on(q4_matmul_f16((q4_half*)xout, (const q4_half*)x, (const q4_half*)w, n, d, batch, x_stride, w_stride, op_stride,
                         w_row_stride, alpha));
}
inline void matmul(half_t* xout, half_t* x, QWeight& w, int inpSize, int opSize, bool accum = false, int loff = -1,
                   int* pPos = nullptr) {                                                                  // :224
    die_on(q4_matmul_q4((q4_half*)xout, (const q4_half*)x, &w, inpSize, opSize, accum ? 1 : 0, loff, pPos));
}
inline void qkv_matvec(half_t* q, half_t* key_cache, half_t* value_cache, half_t* x, QWeight& qw, QWeight& kw, QWeight& vw,
                       int inpSize, int opSize, int loff, int* pPos) {                                     // :235
    die_on(q4_qkv_matvec((q4_half*)q, (q4_half*)key_cache, (q4_half*)value_cache, (const q4_half*)x, &qw, &kw, &vw, inpSize,
                         opSize, loff, pPos));
}
inline void ffn_matvec_silu(half_t* xout, half_t* x, QWeight& gate_w, QWeight& up_w, int inpSize, int opSize) {   // :250
    die_on(q4_ffn_matvec_silu((q4_half*)xout, (const q4_half*)x, &gate_w, &up_w, inpSize, opSize));
}
inline void RoPERotation(half_t* q, half_t* k, int num_heads, int num_kv_heads, int head_size, int* pPos, int loff,
                         float rope_theta) {                                                               // :263
    die_on(q4_rope_rotation((q4_half*)q, (q4_half*)k, num_heads, num_kv_heads, head_size, pPos, loff, rope_theta));
}
inline void MultiHeadAttention(half_t* output, half_t* q, half_t* key_cache, half_t* value_cache, half_t* att, int num_heads,
                               int head_size, int kv_mul, int max_seq_len, int* pPos) {                    // :267
    die_on(q4_multi_head_attention((q4_half*)output, (const q4_half*)q, (const q4_half*)key_cache, (const q4_half*)value_cache,
                                   (q4_half*)att, num_heads, head_size, kv_mul, max_seq_len, pPos));
}
inline void run_llama_network(int* pPos, Config* p, RunState* s, TransformerWeights* w, int seq_len_bin) {   // :286
    die_on(q4_run_llama_network(pPos, p, s, w, seq_len_bin));
}
inline void run_transformer(bool gen_token, Config* p, RunState* s, TransformerWeights* w, bool copyLogits, Sampler* pSampler) {   // :346
    die_on(q4_run_transformer(gen_token ? 1 : 0, p, s, w, copyLogits ? 1 : 0, pSampler));
}
inline void sample(Sampler* sampler, RunState* s, bool gen_token) { die_on(q4_sample(sampler, s, gen_token ? 1 : 0)); }   // sampler.h:43
inline void build_transformer(Transformer* t, char* checkpoint_path, bool perplexity) {                    // :408
    int rc = q4_build_transformer(t, checkpoint_path, perplexity ? 1 : 0);
    if (rc) exit(rc == Q4_ERR_IO ? 1 : EXIT_FAILURE);
}
inline void free_transformer(Transformer* t) { q4_free_transformer(t); }                                    // :428
}  // namespace llama2_q4
