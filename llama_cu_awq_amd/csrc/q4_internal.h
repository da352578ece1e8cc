// q4_internal.h -- shared state and helpers of libllama2_q4.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include "llama2_q4.h"

namespace q4 {

// replaces the reference's file-scope `cudaStream_t stream` (llama2_q4.cu:207)
extern hipStream_t g_stream;
extern int g_fusion;       // 1: fused sequence, 0: 1:1 reference kernel sequence
extern int g_use_graphs;   // USE_CUDA_GRAPHS, llama2_q4.cu:33
extern int g_quiet;
extern char g_last_error[512];
// when non-null every launch carries dispatch timestamps (q4_bench_kernel): hipExtLaunchKernelGGL stamps the
// kernel's own begin/end, so the elapsed time excludes the inter-kernel gap, like rocprofv3's kernel trace
extern hipEvent_t g_ev_start, g_ev_stop;

#define Q4_LAUNCH(kernel, grid, block, smem, ...)                                                              \
    do {                                                                                                       \
        if (q4::g_ev_start)                                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, q4::g_stream, q4::g_ev_start, q4::g_ev_stop, 0,   \
                                  __VA_ARGS__);                                                                \
        else                                                                                                   \
            hipLaunchKernelGGL(kernel, grid, block, smem, q4::g_stream, __VA_ARGS__);                          \
    } while (0)

int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define Q4_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return q4::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define Q4_LAUNCH_CHECK() Q4_HIP(hipGetLastError())

static inline int divUp(int a, int b) { return (a - 1) / b + 1; }   // common.h:80-82

// bytes of RunState::att: n_heads * max(seq_len, dim) halves (the reference's n_heads * dim, llama2_q4.cu:46, overflows for
// seq_len > dim), and at least the flash-decode partial records of the split-context attention at 128 positions per block, eight
// per head or more (one record of head_size + 4 floats per head and chunk; this build keeps no scores there)
static inline size_t att_buffer_bytes(const Config* p) {
    const size_t ref = (size_t)p->n_heads * (p->seq_len > p->dim ? p->seq_len : p->dim) * sizeof(q4_half);
    const int chunks = divUp(p->seq_len, 128) > 8 ? divUp(p->seq_len, 128) : 8;   // (bin 512 works with eight 64-position chunks)
    const size_t rec = (size_t)p->n_heads * chunks * (size_t)(p->dim / p->n_heads + 4) * 8;   // {float, tag} granules
    return ref > rec ? ref : rec;
}

// packed geometry of one QWeight column (llama2_q4.cu:82-98, 227-229)
struct QGeom {
    int K;        // inputElements
    int N;        // opElements
    int pw4;      // uint4 (32 weights) per column = packed_weights_height / 4
    int pzh;      // packed_zeros_height
    int sh;       // scales_height
    int nslots;   // ceil(pw4 / 64): wave-wide uint4 loads per column
};
static inline QGeom make_geom(int K, int N) {
    QGeom g;
    g.K = K; g.N = N;
    g.pw4 = divUp(K, 32);
    g.sh = divUp(K, Q4_GROUP_SIZE);
    g.pzh = divUp(g.sh, 8);
    g.nslots = divUp(g.pw4, 64);
    return g;
}

// ---- internal launchers shared by the API layer and the network (all enqueue on g_stream) ----
struct GemvTune { int cols; int waves; int early; };   // columns per wave, waves per block, early-bird wave slots per SIMD

// fused layer kernels (validated against the unfused chain in tests/test_forward_gpu.py::test_fused_equals_unfused_bits)
// bump: epoch word of the attention -> o-proj launch that follows in the stream (advanced once by this launch), or null
int launch_qkv_fused(q4_half* q, q4_half* kc, q4_half* vc, const q4_half* x, const q4_half* rms_w,
                     const QWeight* qw, const QWeight* kw, const QWeight* vw, int dim, int kv_dim, long long loff,
                     const int* pPos, int head_size, float rope_theta, const float2* rope_table, unsigned* bump);
const float2* rope_table_of(const RunState* s);   // this model's table (q4_runtime.hip), null if none
int rope_table_build(float2** out, int seq_len, int head_size, float theta);   // (cos,sin) table for the fused QKV epilogue
int launch_attention(q4_half* output, const q4_half* q, const q4_half* key_cache, const q4_half* value_cache,
                     int num_heads, int head_size, int kv_mul, int max_seq_len, const int* pPos, float* scratch,
                     size_t scratch_bytes, unsigned* arrive);   // arrive: n_heads zeroed counters (split-context merge by the last block) or null
// attention -> o-proj as one launch (layer_attn.h). Hand-off words of a model (unsigned), zeroed at build and by
// q4_reset_sequence: [SYNC_ERROR] flag of a timed-out wait and [SYNC_EPOCH] launch epoch (advanced by the fused QKV launch) in
// ONE aligned 8-byte word, [SYNC_ARRIVE .. +n_heads) arrival counters of the split-context merge, [SYNC_GRANULES ..) dim/2
// granules {two halves, tag}
enum { SYNC_ERROR = 0, SYNC_EPOCH = 1, SYNC_ARRIVE = 128, SYNC_MAX_HEADS = 512, SYNC_GRANULES = 1024 };
extern int g_ao_mute;
size_t attention_sync_words(int dim);
void attention_oproj_forget_stream(hipStream_t s);
int stream_cu_count();       // CUs the launch stream may use (all of the device, or the bits of its CU mask)
int attention_oproj_form(int dim, int kv_dim, int head_size, int n_heads, int seq_len_bin, bool have_scratch, size_t scratch_bytes,
                         int split_min, int split_chunk);   // >= 0: the fused attention + o-proj launch covers this case
int launch_attention_oproj(q4_half* x, q4_half* xb, const q4_half* q, const q4_half* key_cache, const q4_half* value_cache,
                           const QWeight* wo, int dim, int kv_dim, int n_heads, const int* pPos, int seq_len_bin, unsigned* sync,
                           float* scratch, size_t scratch_bytes, int split_min, int split_chunk);
void attention_set_kv_price(const unsigned* sync, double ticks_per_pos);   // layer_attn.hip: what the split-context launch's hold-back is priced from
double attention_get_kv_price(const unsigned* sync);
extern int g_att_chunk;
extern int g_att_split_min;
extern int g_ao_guard;
extern int g_ao_vslice;
extern int g_ao_hold_pct;
extern int g_ao16;
extern int g_multi_steps;
// Which form of the int4 GEMV launches run. The shipped library only ever holds GEMV_PRODUCT; the profiling library's knob 11
// (q4_set_gemv_early(11, form)) sets the others for A/B runs and for the bit-equality cases of tests/prof_cases.py.
enum GemvForm {
    GEMV_WAVE_OWNED = -1,             // gemv_q4_kernel everywhere (no strips)
    GEMV_PRODUCT = 0,                 // strips where they measure faster (gemv_strip.h, gemv_strip_down.h, gemv_strip_cls.h), wave-owned kernels elsewhere
    GEMV_STRIPS_EVERYWHERE = 8,       // every strips form wherever its shape is covered
    GEMV_STRIPS_D4 = 9, GEMV_STRIPS_D8 = 10,                              // exp/ffn_strip_variants.h: ring depth 4 / 8
    GEMV_STRIPS_PACED_D4 = 12, GEMV_STRIPS_PACED_D8 = 13, GEMV_STRIPS_PACED_D8_ROTATED = 14,   // ... two pieces in flight whatever the depth
    GEMV_PRODUCT_NO_DOWN_STRIPS = 15, // the product's gate/up and classifier choices with the K-split kernel for every down projection
    GEMV_K5120_COLUMN_UNITS = 19,     // the product with K = 5120 gate/up strips on column units instead of pair units
};
extern int g_gemv_form;     // gemv_ffn_strip.hip
// Laboratory hooks: all null in libllama2_q4.so. libllama2_q4_prof.so also links csrc/exp/lab.hip, whose static initialiser registers the
// experiment forms that are kept for A/B (gate/up strips variants and their stamped build, gate/up ablations); the product's dispatcher asks a
// registered hook first. No experiment code is compiled into, or reachable from, the shipped library. (Round 6 removed the forms with a conclusive
// negative -- loader / consumer engine, q/k/v strips, the sampler-epilogue classifier, K / V rings: EXPERIMENTS.md keeps their records, git their code.)
struct GemvArgs;
struct LabHooks {
    bool (*ffn_covers)(const GemvArgs& a);                                        // gate/up: a laboratory form takes this launch
    int (*ffn_launch)(const GemvArgs& a, int waves);
};
extern LabHooks g_lab;      // gemv_ffn_strip.hip
static inline size_t ffn_pair_sync_offset(int dim) { return (attention_sync_words(dim) + 1) & ~(size_t)1; }   // 8-byte aligned (granules), behind the attention words
int classifier_with_final_norm(q4_half* logits, q4_half* x, const q4_half* rms_w, const q4_half* wcls, int dim, int vocab);   // q4_kernels.hip
// (One host thread drives the library, like the reference's: the opt-in map, the prepared-device words of the *_prepare functions, the graph sets and the
// stream caches are plain statics.)
// Opt `kernel` in to `bytes` of dynamic LDS (more than 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize), once per (kernel, device):
// a process-wide flag would leave a second device's kernels at 64 KiB after q4_set_device. Not a stream operation: outside any capture.
int lds_opt_in(const void* kernel, size_t bytes);
int cls_strip_prepare();    // q4_kernels.hip (gemv_strip_cls.h): the same for the classifier's strips kernel
int down_strip_prepare();   // gemv_plain.hip (gemv_strip_down.h): LDS opt-in of the 13B down projection's strips kernel, outside any stream capture
extern unsigned long long* g_dbg;   // profiling build: device buffer for time stamps (q4_set_debug_buffer)
int launch_argmax_feed(const q4_half* x, int size, int* result, volatile int* pPos, int* pPosGpu, q4_half* x_next, const q4_half* table, int dim);
int launch_ffn_fused(q4_half* out, const q4_half* x, const q4_half* rms_w, const QWeight* gate, const QWeight* up,
                     int dim, int hidden);
// rmsnorm + gate/up + SiLU + down projection + residual as ONE launch (gemv_ffn_pair.h; fusion level 4). Its granules lie behind the model's other
// hand-off words, at word ffn_pair_sync_offset(dim)
bool ffn_pair_covers(int dim, int hidden);
size_t ffn_pair_sync_words(int dim, int hidden);
// ... and (fusion level 5) the next layer's rmsnorm + q/k/v + RoPE + KV write as the launch's third phase
struct FfnQkvNext {
    const q4_half* rms_w; const QWeight* wq; const QWeight* wk; const QWeight* wv;
    q4_half* q; q4_half* kc; q4_half* vc;        // RunState::q; the next layer's cache rows (layer offset applied)
    const int* pPos; const float2* rope_table; unsigned* bump; int kv_dim, head_size;
};
bool ffn_qkv_covers(int dim, int hidden, int kv_dim, int head_size, bool have_rope_table);
// ... and (fusion level 6) THIS layer's attention and output projection in front of it: the launch is the whole layer behind its q / k / v
struct FfnLayerAtt {
    q4_half* xb; const q4_half* q; const q4_half* kc; const q4_half* vc;     // RunState::xb, RunState::q; this layer's cache rows (layer offset applied)
    const QWeight* wo; const int* pPos; int n_heads, kv_dim, seq_len_bin;
};
bool layer_att_covers(int dim, int hidden, int kv_dim, int n_heads, int seq_len_bin);
int ffn_pair_prepare();     // gemv_ffn_pair.hip: LDS opt-in, outside any stream capture
int launch_ffn_pair(q4_half* x, q4_half* hb, const q4_half* rms_w, const QWeight* gate, const QWeight* up, const QWeight* down, int dim, int hidden,
                    unsigned* sync, size_t gran_word, unsigned tag_add = 0, const FfnQkvNext* next = nullptr, const FfnLayerAtt* att = nullptr);
unsigned* sync_words_of_state(const RunState* s);   // q4_runtime.hip: the model's hand-off words, or null
extern int g_fp_pre, g_fp_mute, g_fp_nt;

}  // namespace q4
