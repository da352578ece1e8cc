/*
 * llama2_q4.h -- C ABI of the MI355X-native llama2_q4 decode path (libllama2_q4.so).
 *
 * The reference (ankan-ban/llama_cu_awq) has no plugin/FFI layer: it is one CUDA translation
 * unit whose host functions launch kernels on a file-scope stream.  The drop-in boundary is
 * therefore source-level: the same structs (common.h:9-78) and the same host entry points
 * (llama2_q4.cu:209-432, sampler.h:15-82), exported here as `extern "C"` functions over plain
 * pointers and sizes.  Differences forced by a C ABI, and nothing else:
 *   - `half*`  -> `q4_half*` (uint16_t bit pattern of an IEEE binary16),
 *   - `QWeight&` -> `const QWeight*`,
 *   - the file-scope `cudaStream_t stream` (llama2_q4.cu:207) -> `q4_set_stream()/q4_get_stream()`,
 *   - `printf + exit(EXIT_FAILURE)` -> a non-zero return code (the C++ wrappers in
 *     llama2_q4.hpp turn it back into the reference's message + exit), see q4_status.
 * Every declaration cites the reference interface it replaces (file:line in /root/reference).
 *
 * All device pointers are HIP device pointers (hipMalloc) unless stated otherwise.  All launchers
 * enqueue on the current q4 stream and return without synchronising, like the reference.
 */
#ifndef LLAMA2_Q4_H
#define LLAMA2_Q4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t q4_half;          /* fp16 bits (reference: CUDA `half`) */
typedef void* q4_stream_t;         /* hipStream_t */

enum { Q4_MAX_SEQ_LEN_SMEM_KERNEL = 8192,      /* common.h:6 */
       Q4_MAX_SEQ_LEN = 128 * 1024,            /* common.h:7 */
       Q4_GROUP_SIZE = 128,                    /* llama2_q4.cu:31 */
       Q4_MAX_GRAPHS = 8 };                    /* llama2_q4.cu:342 */

typedef enum {
    Q4_OK = 0,
    Q4_ERR_UNSUPPORTED_SIZE = 1,   /* "Unsupported matmul size. Exiting" llama2_q4.cu:215,225,236,251 */
    Q4_ERR_ALLOC = 2,              /* "malloc failed..." llama2_q4.cu:53-58,62-65,129-133 */
    Q4_ERR_IO = 3,                 /* "Couldn't open file" / "Invalid header size" / "error reading weights" :158,412,414 */
    Q4_ERR_HIP = 4,                /* a HIP runtime call failed (the reference never checks) */
    Q4_ERR_ARG = 5
} q4_status;

const char* q4_status_string(int status);
const char* q4_last_error(void);   /* text of the last failing HIP call, "" if none */

/* ---- common.h:9-18 -- also the 32-byte file header, fread raw (llama2_q4.cu:414) ---------- */
typedef struct {
    int dim;          /* transformer dimension */
    int hidden_dim;   /* for ffn layers */
    int n_layers;
    int n_heads;      /* number of query heads */
    int n_kv_heads;   /* number of key/value heads */
    int vocab_size;
    int seq_len;      /* max sequence length */
    float rope_theta;
} Config;

/* ---- common.h:20-24.  Column-major per output column n, K = input length:
 *   weight[n*(K/8) + k/8]  nibble k%8 (LSB first) = q in [0,15]
 *   zeros [n*pzh   + g/8]  nibble g%8, g = k/128, pzh = divUp(divUp(K,128),8)
 *   scales[n*G     + g]    fp16,     G = divUp(K,128)                       (llama2_q4.cu:82-98) */
typedef struct {
    uint32_t* weight;
    uint32_t* zeros;
    q4_half* scales;
} QWeight;

/* common.h:26-36 */
typedef struct {
    q4_half* rms_att_weight;
    q4_half* rms_ffn_weight;
    QWeight wq_q, wq_k, wq_v, wq_o, wq_gate, wq_up, wq_down;
} PerLayerWeight;

/* common.h:38-48 */
typedef struct {
    q4_half* token_embedding_table;   /* (vocab_size, dim) */
    q4_half* wcls;                    /* (vocab_size, dim), not quantised */
    q4_half* rms_final_weight;        /* (dim,) */
    PerLayerWeight* layers;           /* host array of device pointers */
    int num_layers;
} TransformerWeights;

/* common.h:51-54 -- pinned, device-mapped host memory (hipHostMalloc) */
typedef struct {
    volatile int pos;
    int tokens[Q4_MAX_SEQ_LEN];
} SharedData;

/* common.h:56-72, same fields in the same order */
typedef struct {
    q4_half* x;            /* (dim,)   fp16 residual stream */
    q4_half* xb;           /* (dim,) */
    q4_half* hb;           /* (hidden_dim,) */
    q4_half* q;            /* (dim,) */
    q4_half* att;          /* (n_heads, seq_len) scratch; this build's attention keeps scores on chip and
                              uses it only for split-context partials */
    q4_half* logits;       /* (vocab_size,) */
    q4_half* key_cache;    /* (layer, seq_len, kv_dim) */
    q4_half* value_cache;  /* (layer, seq_len, kv_dim) */
    int* pos;              /* device copy of the current position */
    SharedData* shared_data;
    float* logits_array;   /* (seq_len, vocab_size) fp32, perplexity mode only */
} RunState;

/* common.h:74-78 */
typedef struct {
    Config config;
    TransformerWeights weights;
    RunState state;
} Transformer;

/* sampler.h:3-13 */
typedef struct {
    int vocab_size;
    int* indices;
    void* tempStorage_scan;
    void* tempStorage_sort;
    size_t temp_storage_bytes_scan;
    size_t temp_storage_bytes_sort;
    float temperature;
    float topp;
    unsigned long long rng_state;
} Sampler;

/* ---- stream (replaces the file-scope `cudaStream_t stream`, llama2_q4.cu:207,700) ---------- */
int q4_set_device(int device);
int q4_stream_create(q4_stream_t* out);          /* cudaStreamCreate, llama2_q4.cu:700 */
/* a stream restricted to the first n_cus compute units (hipExtStreamCreateWithCUMask): replicas side by side on one GPU.
 * Fusion level 3 counts the waiting blocks of ITS launch against the CUs of ITS stream (the forward-progress guard): models that
 * decode concurrently on one GPU at level 3 must use streams with DISJOINT CU masks (or level 1); two unmasked streams can fill
 * every CU with each other's waiting blocks, which ends as bounded time-outs and a drop to level 1, never as a hang. */
int q4_stream_create_masked(q4_stream_t* out, int n_cus);
int q4_stream_destroy(q4_stream_t s);
void q4_set_stream(q4_stream_t s);
q4_stream_t q4_get_stream(void);
int q4_stream_synchronize(void);                 /* cudaStreamSynchronize(stream), llama2_q4.cu:468 */
int q4_device_synchronize(void);                 /* cudaDeviceSynchronize, perplexity.h:81 */

/* ---- device memory helpers for callers that own buffers (tests, benches, other hosts) -------- */
int q4_malloc(void** dptr, size_t bytes);
int q4_free(void* dptr);
int q4_memcpy_h2d(void* dst, const void* src, size_t bytes);
int q4_memcpy_d2h(void* dst, const void* src, size_t bytes);
int q4_memset(void* dst, int value, size_t bytes);

/* ---- device kernels' launchers (llama2_q4.cu:209-284) ------------------------------------- */

/* rmsnorm(half* o, half* x, half* weight, int size)  llama2_q4.cu:209-212, kernel gpu_kernels.h:72-105 */
int q4_rmsnorm(q4_half* o, const q4_half* x, const q4_half* weight, int size);

/* matmul(half* xout, half* x, half* w, int n, int d, int batch, int x_stride, int w_stride, int op_stride,
 *        int w_row_stride, float alpha)  llama2_q4.cu:214-222, kernel mat_vec_kernel gpu_kernels.h:109-139.
 * fp16 GEMV: xout[b][i] = alpha * sum_j w[b*w_stride + i*w_row_stride + j] * x[b*x_stride + j].
 * w_row_stride == -1 means n. */
int q4_matmul_f16(q4_half* xout, const q4_half* x, const q4_half* w, int n, int d, int batch, int x_stride,
                  int w_stride, int op_stride, int w_row_stride, float alpha);

/* matmul(half* xout, half* x, QWeight& w, int inpSize, int opSize, bool accum, int loff, int* pPos)
 * llama2_q4.cu:224-233, kernel mat_vec_kernel_int4 gpu_kernels.h:213-240 (+ get_mat_vec_int4 :171-210).
 * accum: xout = half(float(xout) + sum).  loff != -1: xout += loff + *pPos * opSize (KV-cache addressing). */
int q4_matmul_q4(q4_half* xout, const q4_half* x, const QWeight* w, int inpSize, int opSize, int accum,
                 int loff, const int* pPos);

/* qkv_matvec(...) llama2_q4.cu:235-248, kernel qkv_matvec_kernel gpu_kernels.h:242-254 */
int q4_qkv_matvec(q4_half* q, q4_half* key_cache, q4_half* value_cache, const q4_half* x, const QWeight* qw,
                  const QWeight* kw, const QWeight* vw, int inpSize, int opSize, int loff, const int* pPos);

/* ffn_matvec_silu(...) llama2_q4.cu:250-261, kernel ffn_matvec_silu_kernel gpu_kernels.h:256-275 */
int q4_ffn_matvec_silu(q4_half* xout, const q4_half* x, const QWeight* gate_w, const QWeight* up_w, int inpSize,
                       int opSize);

/* RoPERotation(half* q, half* k, int num_heads, int num_kv_heads, int head_size, int* pPos, int loff,
 *              float rope_theta)  llama2_q4.cu:263-265, kernel gpu_kernels.h:332-355.  k = key cache base. */
int q4_rope_rotation(q4_half* q, q4_half* k, int num_heads, int num_kv_heads, int head_size, const int* pPos,
                     int loff, float rope_theta);

/* MultiHeadAttention(half* output, half* q, half* key_cache, half* value_cache, half* att, int num_heads,
 *                    int head_size, int kv_mul, int max_seq_len, int* pPos)  llama2_q4.cu:267-284
 * (kernels mat_vec_kernel_simple :142-168, softmax_kernel :357-446, vec_mat_kernel :279-329).
 * key_cache/value_cache already offset by loff.  One flash-decode kernel here: scores never leave the CU. */
int q4_multi_head_attention(q4_half* output, const q4_half* q, const q4_half* key_cache,
                            const q4_half* value_cache, q4_half* att, int num_heads, int head_size, int kv_mul,
                            int max_seq_len, const int* pPos);

/* copy_embedding_kernel gpu_kernels.h:61-69, launch llama2_q4.cu:294. tokens: device-visible int array */
int q4_copy_embedding(q4_half* x, const q4_half* table, int size, const int* tokens, const int* pPos);

/* convert_fp16_to_fp32 gpu_kernels.h:55-59, launch llama2_q4.cu:381 */
int q4_convert_fp16_to_fp32(float* out, const q4_half* in, int elements);

/* argmax_kernel gpu_kernels.h:448-493 (launch sampler.h:49). result: token ring (device-visible),
 * pPos: host-visible position (SharedData::pos), pPosGpu: device position. Ties -> lowest index. */
int q4_argmax(const q4_half* x, int size, int* result, volatile int* pPos, int* pPosGpu, int write_token);

/* ---- network + per-token step ------------------------------------------------------------- */

/* run_llama_network(int* pPos, Config*, RunState*, TransformerWeights*, int seq_len_bin) llama2_q4.cu:286-340 */
int q4_run_llama_network(const int* pPos, const Config* p, RunState* s, const TransformerWeights* w,
                         int seq_len_bin);

/* run_transformer(bool gen_token, Config*, RunState*, TransformerWeights*, bool copyLogits, Sampler*)
 * llama2_q4.cu:346-395: graph bin select / capture-once / replay, optional fp32 logits copy, sample(). */
int q4_run_transformer(int gen_token, const Config* p, RunState* s, const TransformerWeights* w, int copyLogits,
                       Sampler* pSampler);

/* run_transformer with the position supplied by the caller (SharedData::pos is not read back), so that step pos+1
 * can be queued while step pos runs; q4_wait_pos spins on SharedData::pos ("unblocks the CPU", gpu_kernels.h:490)
 * until the device has published position >= pos. generate() uses the pair instead of cudaStreamSynchronize
 * + run_transformer (llama2_q4.cu:468-470): same device order, no idle gap between tokens. */
int q4_run_transformer_at(int pos, int gen_token, const Config* p, RunState* s, const TransformerWeights* w,
                          int copyLogits, Sampler* pSampler);
int q4_wait_pos(const RunState* s, int pos);
/* Greedy steps need nothing from the host between tokens (the device keeps the position and feeds itself the ring), so a
 * token loop may queue Q4_MULTI_STEPS of them as ONE graph replay (sampled steps too: topp_sample_kernel is part of the graph and
 * takes its coin from a device ring by position, filled by the host from the sampler's xorshift stream before the replay): q4_steps_that_fit says how many steps (Q4_MULTI_STEPS or
 * 1) may go out at `pos` -- same gen_token for the whole group, one sequence-length bin, within `steps` --, and
 * q4_run_transformer_steps queues them (nsteps = 1 is q4_run_transformer_at). generate() uses the pair. */
enum { Q4_MULTI_STEPS = 8 };
int q4_steps_that_fit(int pos, int num_prompt_tokens, int steps, const Config* p, const Sampler* sampler);
int q4_run_transformer_steps(int pos, int nsteps, int gen_token, const Config* p, RunState* s, const TransformerWeights* w,
                             int copyLogits, Sampler* pSampler);

/* 0: 1:1 kernel sequence of the reference (10 launches/layer); 1: fused kernels (rmsnorm folded into the consumer GEMV,
 * RoPE + KV write in the QKV epilogue: 5 launches/layer); 3: additionally attention -> o-proj (llama2_q4.cu:320-323)
 * as ONE launch where the geometry has that form (heads of 64 / 128 / 256, multi-head or grouped-query) and every block of the
 * launch fits the stream's CUs at once: the o-proj blocks pull their weights while the heads work and take the heads' output
 * inside the launch (bounded waits, q4_handoff_status), 4 launches/layer. 2 selects 1 (round 2's QKV -> attention -> o-proj
 * launch was measured slower than level 1 and removed). Levels 1 and 3 run the same arithmetic with another fp32 grouping of an
 * attention output's positions (below bin 512 the fused launch's attention role takes 16 positions per wave instruction of a
 * 64-byte V slice, the stand-alone kernel 4; 8 waves instead of 16): logits agree within the model's tolerance in every bin, not
 * bit for bit. After a timed-out hand-off the library runs level 1 for the next 16 sequences (32, 64, ... after further
 * time-outs), then tries the level it had again; q4_set_fusion(level) re-arms it at once. Resets captured graphs.
 * 4: additionally the FFN half of a layer (rmsnorm + gate/up + SiLU + down projection + residual add, llama2_q4.cu:326-332) as ONE
 * launch where q4_ffn_pair_covers says so (Llama-2-7B's dim / hidden_dim): one block per CU computes its slice of hb, hands it to every
 * other CU inside the launch and multiplies its columns of the down projection, whose weights it has meanwhile streamed into LDS;
 * 3 launches/layer, bit-identical to levels 1 / 3 (same bounded waits, same fallback after a time-out).
 * 5 (default): that launch also runs rmsnorm + q/k/v + RoPE + KV write of the NEXT layer (llama2_q4.cu:300-317) as its third phase, on
 * weights it streams into LDS while it multiplies the down projection -- multi-head models of Llama-2-7B's shape; 2 launches/layer (the
 * first layer keeps its own QKV launch, the last layer's FFN half runs as at level 4), bit-identical as well.
 * 6 (opt-in; measured 1-2 % SLOWER than 5): below the split-context bins that launch also begins with THIS layer's attention and
 * output projection (llama2_q4.cu:320-323) -- half of its blocks run the attention role's (head, V slice) units, the other half the output
 * projection; the whole layer behind its q / k / v is ONE launch, 1 launch/layer; bit-identical to levels 3 / 4 / 5. */
void q4_set_fusion(int level);
int q4_get_fusion(void);
/* 1: at fusion levels 4 / 5 a layer's FFN half of these sizes runs as one launch on the current device and stream (csrc/gemv_ffn_pair.h) */
int q4_ffn_pair_covers(int dim, int hidden_dim);
/* 1 (default): hipGraph capture/replay as USE_CUDA_GRAPHS llama2_q4.cu:33; 0: eager launches with the exact context length (the
 * reference's other path, :374); 2: eager launches with the graph path's sequence-length bin -- exactly what the graphs run, one launch
 * at a time (the mode to profile in: rocprofv3 cannot trace inside a graph capture) */
void q4_set_use_graphs(int enable);
void q4_reset_graphs(void);   /* drop captured graphs (main() cleanup llama2_q4.cu:713-716) */
/* Graphs captured so far in this process. Captured graphs are kept per model (RunState), for up to four live models: a host that alternates
 * them replays each one's graphs; beyond four the least recently used model's graphs are dropped and captured again on its next turn --
 * this counter is how a host sees that happen. */
int q4_graph_captures(void);

/* build_sampler / destroy_sampler sampler.h:15-29; random_u32 / random_f32 :31-40; sample :43-82 */
int build_sampler(Sampler* sampler, int vocab_size, float temperature, float topp, unsigned long long rng_seed);
void destroy_sampler(Sampler* sampler);
unsigned int random_u32(unsigned long long* state);
float random_f32(unsigned long long* state);
int q4_sample(Sampler* sampler, RunState* s, int gen_token);

/* build_transformer(Transformer*, char* checkpoint_path, bool perplexity) llama2_q4.cu:408-426 (prints the
 * same "Model params" / "Loading Weights... done!" lines unless quiet), free_transformer :428-432 */
int q4_build_transformer(Transformer* t, const char* checkpoint_path, int perplexity);
void q4_free_transformer(Transformer* t);
void q4_set_quiet(int quiet);

/* opaque-handle convenience for FFI hosts that cannot lay out the structs (ctypes, cgo, JNI) */
Transformer* q4_transformer_new(const char* checkpoint_path, int perplexity, int* status);
void q4_transformer_delete(Transformer* t);
const Config* q4_transformer_config(const Transformer* t);
RunState* q4_transformer_state(Transformer* t);
TransformerWeights* q4_transformer_weights(Transformer* t);
Sampler* q4_sampler_new(int vocab_size, float temperature, float topp, unsigned long long rng_seed);
void q4_sampler_delete(Sampler* s);
/* generate()'s state reset, llama2_q4.cu:461-463: pos = 0, copy prompt tokens into the shared ring */
int q4_reset_sequence(RunState* s, const int* prompt_tokens, int num_prompt_tokens);
int q4_shared_pos(const RunState* s);
/* Fusion level 3 waits inside a launch with BOUNDED spins; one that ran out sets a device flag and the results from then on
 * are invalid. Synchronises the stream; Q4_OK, or -- once per incident -- Q4_ERR_HIP with q4_last_error() set, after which
 * the hand-off state is cleared and the library runs at fusion level 1 (no in-launch waits): redo the sequence
 * (q4_reset_sequence). q4_generate / q4_generate_ids / q4_perplexity_ids do that themselves, q4_chat reports and stops.
 * A loop built on q4_run_transformer should call it wherever it synchronises. q4_handoff_timeouts: incidents so far. */
int q4_handoff_status(const RunState* s);
int q4_handoff_timeouts(void);
/* Diagnostic: duration of the K / V stream per context position, in ticks of 10 ns, as q4_build_transformer measured it for this model on this device
 * (the split-context attention -> o-proj launch prices its held-back weight requests from it); 0: not measured (seq_len < 1024). */
double q4_kv_stream_price(const RunState* s);
int q4_shared_token(const RunState* s, int index);
/* parity dumps: synchronise, then copy fp16 logits / a KV row / the residual to the host */
int q4_get_logits(const Transformer* t, q4_half* host_out);
int q4_get_kv_row(const Transformer* t, int layer, int pos, q4_half* host_k, q4_half* host_v);
int q4_get_logits_array(const Transformer* t, int num_pos, float* host_out);

/* ---- host drivers (llama2_q4.cu:436-601, perplexity.h) ---------------------------------------- */
struct Tokenizer;
/* generate(Transformer*, Tokenizer*, Sampler*, char* prompt, int steps) llama2_q4.cu:436-492.
 * Returns the achieved tok/s it printed; timed_tokens/seconds optionally returned. */
double q4_generate(Transformer* t, struct Tokenizer* tokenizer, Sampler* sampler, const char* prompt, int steps,
                   int* timed_tokens, double* seconds);
/* token-id variant used by benches/tests when no tokenizer file is present: same loop, same timing rule,
 * no printing. out_tokens (steps+1 ints, may be NULL) receives the token ring. Returns tokens/s. */
double q4_generate_ids(Transformer* t, Sampler* sampler, const int* prompt_tokens, int num_prompt_tokens,
                       int steps, int* out_tokens, int* timed_tokens, double* seconds);
/* chat(...) llama2_q4.cu:507-601 */
void q4_chat(Transformer* t, struct Tokenizer* tokenizer, Sampler* sampler, const char* cli_user_prompt,
             const char* cli_system_prompt, int steps);
/* softmax / compute_perplexity perplexity.h:3-51 (host math) */
void q4_softmax_f32(float* x, int size);
float compute_perplexity(const int* tokens, float* logits, int num_tokens, int vocab_size);
/* get_dataset_perplexity perplexity.h:57-97, parseDataSetAndComputePreplexity :99-139 */
float q4_get_dataset_perplexity(char* dataset, struct Tokenizer* tokenizer, Transformer* t, Sampler* sampler);
double q4_parse_dataset_and_compute_perplexity(const char* textFileName, struct Tokenizer* tokenizer,
                                               Transformer* t, Sampler* sampler);
/* teacher-forced logits for given token ids (perplexity path without a tokenizer): runs num_tokens steps with
 * copyLogits, returns perplexity of targets[i] under step i's logits */
float q4_perplexity_ids(Transformer* t, Sampler* sampler, const int* tokens_with_bos, int num_tokens);

/* ---- tokenizer.h:1-223 ------------------------------------------------------------------------ */
struct Tokenizer* q4_tokenizer_new(const char* tokenizer_path, int vocab_size);   /* build_tokenizer :35-59 */
void q4_tokenizer_delete(struct Tokenizer* t);                                     /* free_tokenizer :61-66 */
/* encode :102-223; tokens must hold strlen(text)+3 ints */
int q4_tokenizer_encode(struct Tokenizer* t, const char* text, int bos, int eos, int* tokens, int* n_tokens);
const char* q4_tokenizer_decode(struct Tokenizer* t, int prev_token, int token);   /* decode :68-79 */
int q4_tokenizer_max_token_length(const struct Tokenizer* t);

/* ---- CLI: main() llama2_q4.cu:622-720 behind a callable (the llama2_q4 executable calls this) --- */
int q4_main(int argc, char** argv);
/* the flag parser alone (llama2_q4.cu:624-690), for tests: fills the struct, returns 0, or 1 where the
 * reference would call error_usage() */
typedef struct {
    const char* checkpoint_path;
    const char* tokenizer_path;
    const char* dataset_path;
    int steps;
    const char* prompt;
    int perplexity;
    float temperature;
    float topp;
    unsigned long long rng_seed;
    const char* mode;
    const char* system_prompt;
    int seed_from_time;
} q4_cli_args;
int q4_parse_args(int argc, char** argv, q4_cli_args* out);

/* ---- measurement helpers (bench.py; not part of the reference surface) ------------------------ */
/* Launch `kernel_id` `iters` times over a ring of `ring` distinct weight sets (defeats the 256 MiB Infinity
 * Cache), each launch bracketed by dispatch timestamps (hipExtLaunchKernel start/stop events on the q4 stream);
 * returns the average pure kernel duration in microseconds, <0 on error. The ring is layers[i % ring] of `w`.
 *   0: fused rmsnorm + gate/up + SiLU (the decode path's dominant kernel)   1: plain int4 GEMV dim->hidden (gate)
 *   2: plain int4 GEMV hidden->dim (down, accum)   3: fused rmsnorm + qkv + rope   4: o-proj (accum)
 *   5: fp16 classifier (ring ignored)  */
double q4_bench_kernel(int kernel_id, const Config* p, RunState* s, const TransformerWeights* w, int iters,
                       double* min_us, double* max_us);
/* Steady-state cost of one launch of kernel_id INSIDE a hipGraph (kernel + boundary, what the decode loop pays):
 * `iters` launches over the ring of layers are captured into one graph and replayed `reps` times; wall-clock
 * microseconds per launch. ids as above, plus 6: attention, 7: rmsnorm, 8: argmax, 9: embedding copy. */
double q4_bench_kernel_graph(int kernel_id, const Config* p, RunState* s, const TransformerWeights* w, int iters,
                             int reps);
/* Average duration of one launch class INSIDE the eager decode network (inputs produced by the previous kernel,
 * caches as the token loop leaves them): runs `tokens` greedy decode steps from the current position with dispatch
 * timestamps on the launches in time_mask | report_mask, statistics over report_mask (1 qkv, 2 attention, 4 o-proj, 8 gate/up, 16 down,
 * 32 final norm + classifier, 64 embedding); microseconds, <0 on error. This is the duration bench.py's roofline
 * uses and the one `rocprofv3 --kernel-trace` reports for `bench.py --no-graphs`. */
double q4_bench_in_network(int time_mask, int report_mask, const Config* p, RunState* s, const TransformerWeights* w,
                           int tokens, double* min_us, double* max_us, int* launches);
int q4_device_info(char* name, int name_len, int* cu_count, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LLAMA2_Q4_H */
