// q4_runtime.hip -- model state, loader, per-token network, hipGraph step, sampler and token loops.
// Mirrors llama2_q4.cu:38-202 (memory + loader), :286-340 (run_llama_network), :342-395 (run_transformer),
// :436-492 (generate), sampler.h, perplexity.h:57-97.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <map>
#include <vector>
#include "q4_internal.h"

namespace q4 {

hipStream_t g_stream = nullptr;
int g_fusion = 5;
int g_multi_steps = Q4_MULTI_STEPS;   // greedy steps per graph replay in the token loops (profiling build: q4_set_gemv_early(7, n))
int g_use_graphs = 1;
int g_quiet = 0;
static int g_rearm_after = 0;          // > 0: sequences left at fusion level 1 before the level in force is tried again (after a timed-out hand-off)
static int g_rearm_level = 5;          // the fusion level that comes back when the probation ends
static int g_rearm_backoff = 16;       // sequences to sit out after the next time-out: doubles every time, so a box that keeps stalling settles at level 1
static int g_handoff_timeouts = 0;     // timed-out in-launch waits seen by q4_handoff_status since the library was loaded
char g_last_error[512] = "";
hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    snprintf(g_last_error, sizeof(g_last_error), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
    fprintf(stderr, "llama2_q4: HIP error %s\n", g_last_error);
    return Q4_ERR_HIP;
}

// every Transformer owns two HBM slabs (weights, run state) instead of ~750 small allocations: layers sit
// back to back in HBM in the order the token loop streams them.
struct Slabs {
    void* weights = nullptr;
    void* state = nullptr;
    void* shared = nullptr;
    void* logits_array = nullptr;
    float2* rope_table = nullptr;   // [seq_len][head_size/2] (cos, sin), lives exactly as long as the Transformer
    unsigned* sync = nullptr;       // hand-off words of the attention -> o-proj launch (layout: q4_internal.h SYNC_*)
};
static std::map<const Transformer*, Slabs> g_slabs;
// the network entry points take (Config, RunState, TransformerWeights), not the Transformer: tables are found by RunState
static std::map<const RunState*, const float2*> g_rope_by_state;
static std::map<const RunState*, unsigned*> g_sync_by_state;
static std::map<const RunState*, size_t> g_sync_words;   // words per model; word 0 is the sticky error flag
static std::map<const RunState*, size_t> g_att_bytes;    // bytes of RunState::att (the split-context records live there)
const float2* rope_table_of(const RunState* s) {
    auto it = g_rope_by_state.find(s);
    return it == g_rope_by_state.end() ? nullptr : it->second;
}
static unsigned* sync_words_of(const RunState* s) {
    auto it = g_sync_by_state.find(s);
    return it == g_sync_by_state.end() ? nullptr : it->second;
}
// Hand-off state between sequences / launch-sequence changes / after a time-out. The EPOCH word is never rewound: the tag of a
// launch is the epoch as it finds it, and the tagged words that outlive a launch -- the attention granules and the split-context
// records in RunState::att -- validate themselves by tag alone, so a tag must never repeat during the life of the model (a second
// sequence that reached the same (position, layer) used to re-create the first one's tag and could merge its stale records).
// Cleared: arrival counters and granules, and the error word when asked. Past 2^31 launches (days of decoding) the epoch does
// start over, together with every buffer that holds tags.
unsigned* sync_words_of_state(const RunState* s) { return sync_words_of(s); }
static int clear_handoff_state(const RunState* s, bool error_too) {
    unsigned* sync = sync_words_of(s);
    auto it = g_sync_words.find(s);
    if (!sync || it == g_sync_words.end() || it->second <= SYNC_EPOCH + 1) return Q4_OK;
    if (error_too) Q4_HIP(hipMemsetAsync(sync + SYNC_ERROR, 0, sizeof(unsigned), g_stream));
    Q4_HIP(hipMemsetAsync(sync + SYNC_EPOCH + 1, 0, (it->second - SYNC_EPOCH - 1) * sizeof(unsigned), g_stream));
    Q4_HIP(hipStreamSynchronize(g_stream));
    unsigned epoch = 0;
    Q4_HIP(hipMemcpy(&epoch, sync + SYNC_EPOCH, sizeof(epoch), hipMemcpyDeviceToHost));
    if (epoch >= 0x80000000u) {
        Q4_HIP(hipMemsetAsync(sync + SYNC_EPOCH, 0, sizeof(unsigned), g_stream));
        auto ab = g_att_bytes.find(s);
        if (ab != g_att_bytes.end() && s->att) Q4_HIP(hipMemsetAsync(s->att, 0, ab->second, g_stream));
        Q4_HIP(hipStreamSynchronize(g_stream));
    }
    return Q4_OK;
}

int lds_opt_in(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return Q4_OK;
    static std::map<std::pair<const void*, int>, size_t> opted;
    int dev = 0;
    Q4_HIP(hipGetDevice(&dev));
    size_t& have = opted[{kernel, dev}];
    if (bytes > have) {
        Q4_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return Q4_OK;
}

// graphs: [bin][variant]; variant bit0 = gen_token, bit1 = copyLogits, bit2 = sampled (the sampler launch, its temperature / top-p / coin ring baked in),
// bit3 = Q4_MULTI_STEPS steps per graph. A captured graph holds one model's pointers, so the sets are kept PER MODEL (its RunState): a host that alternates
// a few models on one GPU replays each one's graphs (llama2_q4.cu:342-344 keeps one set for its one model); beyond GRAPH_OWNERS live models the least
// recently used set is dropped and captured again on its next turn (q4_graph_captures counts: a host can see it happen).
enum { GRAPH_OWNERS = 4 };
struct GraphSet {
    const void* owner;
    unsigned long long used;
    hipGraphExec_t exec[Q4_MAX_GRAPHS][16];
    bool captured[Q4_MAX_GRAPHS][16];
    const Sampler* sampler;            // what the set's sampled graphs have baked in
    float temperature, topp;
    const float* coins;
};
static GraphSet g_sets[GRAPH_OWNERS];
static unsigned long long g_set_clock = 0;
static int g_graph_captures = 0;
static void drop_graphs(GraphSet& gs, bool sampled_only) {
    for (int i = 0; i < Q4_MAX_GRAPHS; i++)
        for (int v = 0; v < 16; v++)
            if (gs.captured[i][v] && (!sampled_only || (v & 4))) {
                hipGraphExecDestroy(gs.exec[i][v]);
                gs.captured[i][v] = false;
            }
    if (!sampled_only) gs.owner = nullptr;
    gs.sampler = nullptr; gs.coins = nullptr;
}
static GraphSet& graph_set_of(const void* owner) {
    GraphSet* pick = nullptr;
    for (GraphSet& gs : g_sets)
        if (gs.owner == owner) { pick = &gs; break; }
    if (!pick) {
        for (GraphSet& gs : g_sets)
            if (!pick || (gs.owner == nullptr && pick->owner != nullptr) || (((gs.owner == nullptr) == (pick->owner == nullptr)) && gs.used < pick->used)) pick = &gs;
        if (pick->owner) drop_graphs(*pick, false);
        pick->owner = owner;
    }
    pick->used = ++g_set_clock;
    return *pick;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t qweight_bytes(int height, int width, size_t* wb, size_t* zb, size_t* sb) {
    // llama2_q4.cu:82-98
    size_t packed_wt_height = (size_t)divUp(height, 32) * 4;
    size_t scales_height = (size_t)divUp(height, Q4_GROUP_SIZE);
    size_t packed_zeros_height = (size_t)divUp((int)scales_height, 8);
    *wb = packed_wt_height * width * sizeof(uint32_t);
    *zb = packed_zeros_height * width * sizeof(uint32_t);
    *sb = scales_height * width * sizeof(q4_half);
    return *wb + *zb + *sb;
}

}  // namespace q4

using namespace q4;

extern "C" __attribute__((visibility("hidden"))) int q4_copy_logits_at_pos(float* logits_array, const q4_half* logits, int vocab_size, const int* pPos);

extern "C" {

const char* q4_status_string(int status) {
    switch (status) {
        case Q4_OK: return "ok";
        case Q4_ERR_UNSUPPORTED_SIZE: return "Unsupported matmul size. Exiting";   // llama2_q4.cu:215
        case Q4_ERR_ALLOC: return "malloc failed!";                                // :131
        case Q4_ERR_IO: return "error reading weights";                            // :158
        case Q4_ERR_HIP: return "HIP runtime error";
        case Q4_ERR_ARG: return "invalid argument";
    }
    return "unknown";
}
const char* q4_last_error(void) { return g_last_error; }

int q4_set_device(int device) {
    Q4_HIP(hipSetDevice(device));
    // LDS opt-ins of kernels the public matmuls may launch: attribute calls, made before any stream capture can begin
    const int rc = down_strip_prepare();
    return rc ? rc : cls_strip_prepare();
}
int q4_stream_create(q4_stream_t* out) {
    hipStream_t s;
    Q4_HIP(hipStreamCreate(&s));
    *out = (q4_stream_t)s;
    return Q4_OK;
}
// A stream restricted to the first `n_cus` compute units (hipExtStreamCreateWithCUMask): lets a host run replicas side by
// side on one GPU, and lets the tests run the in-launch hand-offs with fewer resident blocks than a whole device offers.
int q4_stream_create_masked(q4_stream_t* out, int n_cus) {
    int dev = 0, total = 0;
    Q4_HIP(hipGetDevice(&dev));
    Q4_HIP(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cus < 1 || n_cus > total) return Q4_ERR_ARG;
    uint32_t mask[16] = {};
    const int words = (total + 31) / 32;
    if (words > 16) return Q4_ERR_ARG;
    // CU bits are interleaved over the XCDs in the mask (bit i = CU i / 8 of XCD i % 8 on gfx950): taking the first n bits
    // spreads the stream over all XCDs
    for (int i = 0; i < n_cus; i++) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s;
    Q4_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    *out = (q4_stream_t)s;
    return Q4_OK;
}
int q4_stream_destroy(q4_stream_t s) {
    if ((hipStream_t)s == g_stream) g_stream = nullptr;
    attention_oproj_forget_stream((hipStream_t)s);
    Q4_HIP(hipStreamDestroy((hipStream_t)s));
    return Q4_OK;
}
void q4_set_stream(q4_stream_t s) { g_stream = (hipStream_t)s; }
q4_stream_t q4_get_stream(void) { return (q4_stream_t)g_stream; }
int q4_stream_synchronize(void) { Q4_HIP(hipStreamSynchronize(g_stream)); return Q4_OK; }
int q4_device_synchronize(void) { Q4_HIP(hipDeviceSynchronize()); return Q4_OK; }

int q4_malloc(void** dptr, size_t bytes) { Q4_HIP(hipMalloc(dptr, bytes)); return Q4_OK; }
int q4_free(void* dptr) { Q4_HIP(hipFree(dptr)); return Q4_OK; }
int q4_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    Q4_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_stream));
    Q4_HIP(hipStreamSynchronize(g_stream));
    return Q4_OK;
}
int q4_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    Q4_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_stream));
    Q4_HIP(hipStreamSynchronize(g_stream));
    return Q4_OK;
}
int q4_memset(void* dst, int value, size_t bytes) {
    Q4_HIP(hipMemsetAsync(dst, value, bytes, g_stream));
    return Q4_OK;
}

// 0: the reference's 1:1 kernel sequence; 1: fused rmsnorm / RoPE / SiLU epilogues (five launches per layer); 3: attention ->
// o-proj as one launch on top of that (default). Level 2 (QKV -> attention -> o-proj as one launch) was measured slower than
// level 1 in round 2 and removed in round 3: the value selects level 1.
void q4_set_fusion(int level) {
    g_fusion = level <= 0 ? 0 : level >= 6 ? 6 : level == 5 ? 5 : level == 4 ? 4 : level == 3 ? 3 : 1;
    g_rearm_after = 0;          // an explicit choice ends the probation after a time-out (and is the documented way to re-arm at once)
    q4_reset_graphs();
    if (g_stream)
        for (auto& kv : g_sync_by_state) (void)clear_handoff_state(kv.first, false);   // no stale counters / granules across a change of launch sequence
}
int q4_get_fusion(void) { return g_fusion; }
int q4_ffn_pair_covers(int dim, int hidden_dim) { return ffn_pair_covers(dim, hidden_dim) ? 1 : 0; }
// 1: captured graphs (USE_CUDA_GRAPHS, llama2_q4.cu:33); 0: eager launches with the exact context length (the reference's other path, :374);
// 2: eager launches with the graph path's sequence-length BIN -- what the graphs run, one launch at a time: the mode to profile in
// (rocprofv3 cannot trace inside a graph capture; with 0 the attention launch picks its form from the context length, not from the bin)
void q4_set_use_graphs(int enable) { g_use_graphs = enable == 2 ? 2 : enable ? 1 : 0; }
void q4_set_quiet(int quiet) { g_quiet = quiet; }

void q4_reset_graphs(void) {
    for (GraphSet& gs : g_sets) drop_graphs(gs, false);
}
int q4_graph_captures(void) { return g_graph_captures; }

int q4_device_info(char* name, int name_len, int* cu_count, size_t* hbm_bytes) {
    int dev = 0;
    Q4_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    Q4_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return Q4_OK;
}

// ---------------------------------------------------------------------------------------------------
// build_transformer: llama2_q4.cu:408-426 = header fread + malloc_weights (:106-134) +
// checkpoint_init_weights (:172-202) + malloc_run_state (:38-67)
static int read_to_device(void* dst, FILE* fp, size_t bytes, void* scratch) {
    if (fread(scratch, 1, bytes, fp) != bytes) {                                   // readWeight :157-160
        printf("error reading weights");
        return Q4_ERR_IO;
    }
    Q4_HIP(hipMemcpy(dst, scratch, bytes, hipMemcpyHostToDevice));
    return Q4_OK;
}

struct SlabCursor {
    char* base;
    size_t off;
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off = align_up(off + bytes, 256);
        return p;
    }
};

static void carve_qweight(SlabCursor& c, QWeight* w, int height, int width) {
    size_t wb, zb, sb;
    qweight_bytes(height, width, &wb, &zb, &sb);
    w->weight = (uint32_t*)c.take(wb);
    w->zeros = (uint32_t*)c.take(zb);
    w->scales = (q4_half*)c.take(sb);
}

static void carve_weights(SlabCursor& c, TransformerWeights* w, const Config* p) {
    const int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    w->token_embedding_table = (q4_half*)c.take((size_t)p->vocab_size * p->dim * sizeof(q4_half));
    w->wcls = (q4_half*)c.take((size_t)p->vocab_size * p->dim * sizeof(q4_half));
    w->rms_final_weight = (q4_half*)c.take((size_t)p->dim * sizeof(q4_half));
    for (int l = 0; l < p->n_layers; l++) {
        PerLayerWeight* layer = &w->layers[l];
        carve_qweight(c, &layer->wq_q, p->dim, p->dim);
        carve_qweight(c, &layer->wq_k, p->dim, kv_dim);
        carve_qweight(c, &layer->wq_v, p->dim, kv_dim);
        carve_qweight(c, &layer->wq_o, p->dim, p->dim);
        carve_qweight(c, &layer->wq_up, p->dim, p->hidden_dim);
        carve_qweight(c, &layer->wq_gate, p->dim, p->hidden_dim);
        carve_qweight(c, &layer->wq_down, p->hidden_dim, p->dim);
        layer->rms_att_weight = (q4_half*)c.take((size_t)p->dim * sizeof(q4_half));
        layer->rms_ffn_weight = (q4_half*)c.take((size_t)p->dim * sizeof(q4_half));
    }
}

static int upload_qweight(QWeight* w, FILE* fp, int height, int width, void* scratch) {
    size_t wb, zb, sb;
    qweight_bytes(height, width, &wb, &zb, &sb);
    int rc;
    if ((rc = read_to_device(w->weight, fp, wb, scratch))) return rc;              // uploadQWeight :162-170
    if ((rc = read_to_device(w->zeros, fp, zb, scratch))) return rc;
    if ((rc = read_to_device(w->scales, fp, sb, scratch))) return rc;
    return Q4_OK;
}

// The split-context attention -> o-proj launch holds its o-proj role's weight requests back until the K / V stream of the context is about to end
// (layer_attn.h). What it needs is the stream's duration per context position ON THIS DEVICE, for THIS model's rows: measured here, once per model, with the
// product's own split-context attention launch at two context lengths (dispatch timestamps on the launch stream; the slope is the price, the fixed part of
// a launch drops out) -- instead of a constant measured on one box in round 5. A few hundred microseconds of build time; the run
// state is all zeros at this point and stays so (the launch writes xb and its scratch records only).
static void measure_kv_price(const Config* p, RunState* s, unsigned* sync) {
    const int head_size = p->dim / p->n_heads;
    const int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    const int n2 = p->seq_len < 2048 ? p->seq_len : 2048, n1 = n2 / 2;
    if (n1 < 512 || p->n_heads > SYNC_MAX_HEADS || !(head_size == 64 || head_size == 128 || head_size == 256) || !s->att || g_stream == nullptr) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return; }
    double t[2] = {0.0, 0.0};
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++) {
        const int n = k ? n2 : n1;
        const int last = n - 1;
        ok = hipMemcpyAsync(s->pos, &last, sizeof(int), hipMemcpyHostToDevice, g_stream) == hipSuccess && hipStreamSynchronize(g_stream) == hipSuccess;
        double best = 1e30;
        for (int rep = 0; rep < 12 && ok; rep++) {       // the minimum of a dozen launches: the stream's own time, not a neighbour's
            // every launch on ANOTHER layer's rows (the token loop comes back to a layer after all the others: its rows come from HBM, while a
            // repeated read of one layer's 33 MB would come out of the 256 MB Infinity Cache at twice the rate)
            const size_t loff = (size_t)((rep * 7 + k * 3) % p->n_layers) * p->seq_len * kv_dim;
            g_ev_start = e0; g_ev_stop = e1;
            const int rc = launch_attention(s->xb, s->q, s->key_cache + loff, s->value_cache + loff, p->n_heads, head_size, p->n_heads / p->n_kv_heads, n, s->pos,
                                            (float*)s->att, att_buffer_bytes(p), p->n_heads <= SYNC_MAX_HEADS ? sync + SYNC_ARRIVE : nullptr);   // ONE launch: merged by each head's last block (the counters re-arm themselves)
            g_ev_start = g_ev_stop = nullptr;
            float ms = 0.f;
            ok = rc == Q4_OK && hipStreamSynchronize(g_stream) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (ok && rep >= 2 && ms > 0.f && ms < best) best = ms;
        }
        t[k] = best;
    }
    const int zero = 0;
    (void)hipMemcpyAsync(s->pos, &zero, sizeof(int), hipMemcpyHostToDevice, g_stream);
    (void)hipMemsetAsync(s->xb, 0, (size_t)p->dim * sizeof(q4_half), g_stream);
    (void)hipStreamSynchronize(g_stream);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (!ok) { (void)hipGetLastError(); if (getenv("Q4_DEBUG_KV")) fprintf(stderr, "llama2_q4: kv price: measurement failed\n"); return; }
    const double ticks_per_pos = (t[1] - t[0]) * 1e-3 / (double)(n2 - n1) * 1e8;      // ms -> s -> 10 ns ticks
    // sanity: between a third of and three times the rows' bytes at 7.6 TB/s (what MI355X measures) -- outside that the measurement was disturbed and
    // the constant stands in
    const double nominal = (4.0 * kv_dim) / 7.6e12 * 1e8;
    if (getenv("Q4_DEBUG_KV")) fprintf(stderr, "llama2_q4: kv price: %d positions %.3f us, %d positions %.3f us -> %.4f ticks / position (nominal %.4f)\n", n1, t[0] * 1e3, n2, t[1] * 1e3, ticks_per_pos, nominal);
    if (ticks_per_pos > nominal / 3.0 && ticks_per_pos < nominal * 3.0) attention_set_kv_price(sync, ticks_per_pos);
}

int q4_build_transformer(Transformer* t, const char* checkpoint_path, int perplexity) {
    memset(t, 0, sizeof(*t));
    FILE* file = fopen(checkpoint_path, "rb");
    if (!file) { printf("Couldn't open file %s\n", checkpoint_path); return Q4_ERR_IO; }   // :412
    if (fread(&t->config, sizeof(Config), 1, file) != 1) { printf("Invalid header size\n"); fclose(file); return Q4_ERR_IO; }
    Config* p = &t->config;
    if (!g_quiet)
        printf("\nModel params:- \ndim: %d \nhidden_dim: %d\nn_heads: %d\nn_kv_heads: %d\nn_layers: %d\nseq_len: %d\nvocab_size: %d\nrope_theta: %g\n",
               p->dim, p->hidden_dim, p->n_heads, p->n_kv_heads, p->n_layers, p->seq_len, p->vocab_size, p->rope_theta);   // :416-417
    if (p->dim <= 0 || p->n_heads <= 0 || p->n_kv_heads <= 0 || p->n_layers <= 0 || p->vocab_size <= 0 || p->seq_len <= 0 ||
        p->dim % p->n_heads || p->n_heads % p->n_kv_heads || (p->dim & 31) || (p->hidden_dim & 31) ||
        p->seq_len > Q4_MAX_SEQ_LEN) {
        printf("Unsupported model geometry\n");
        fclose(file);
        return Q4_ERR_UNSUPPORTED_SIZE;
    }
    const int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    Slabs slabs;
    TransformerWeights* w = &t->weights;
    w->layers = (PerLayerWeight*)calloc(p->n_layers, sizeof(PerLayerWeight));      // malloc_weights :109
    w->num_layers = p->n_layers;
    if (!w->layers) { printf("malloc failed!\n"); fclose(file); return Q4_ERR_ALLOC; }

    // pass 1 sizes the weight slab, pass 2 hands out pointers
    SlabCursor dry = {nullptr, 0};
    carve_weights(dry, w, p);
    if (hipMalloc(&slabs.weights, dry.off) != hipSuccess) { printf("malloc failed!\n"); fclose(file); return Q4_ERR_ALLOC; }
    SlabCursor cur = {(char*)slabs.weights, 0};
    carve_weights(cur, w, p);

    // checkpoint_init_weights :172-202 (same tensor order: embedding, wcls, final norm; per layer q,k,v,o,up,gate,down,norms)
    // staging buffer = the largest single read (the reference sizes it max(vocab, hidden) * dim halves, :176, which a q/o
    // matrix outgrows when dim > 4 * max(vocab, hidden))
    size_t scratch_size = (size_t)(p->vocab_size > p->hidden_dim ? p->vocab_size : p->hidden_dim) * p->dim * sizeof(q4_half);
    {
        size_t wb, zb, sb;
        qweight_bytes(p->dim, p->dim, &wb, &zb, &sb);
        if (wb > scratch_size) scratch_size = wb;
        qweight_bytes(p->hidden_dim, p->dim, &wb, &zb, &sb);
        if (wb > scratch_size) scratch_size = wb;
    }
    void* scratch = nullptr;
    if (hipHostMalloc(&scratch, scratch_size, hipHostMallocDefault) != hipSuccess) scratch = nullptr;
    bool scratch_pinned = scratch != nullptr;
    if (!scratch) scratch = malloc(scratch_size);
    if (!g_quiet) printf("\nLoading Weights... ");
    int rc = Q4_OK;
    do {
        if ((rc = read_to_device(w->token_embedding_table, file, (size_t)p->vocab_size * p->dim * sizeof(q4_half), scratch))) break;
        if ((rc = read_to_device(w->wcls, file, (size_t)p->vocab_size * p->dim * sizeof(q4_half), scratch))) break;
        if ((rc = read_to_device(w->rms_final_weight, file, (size_t)p->dim * sizeof(q4_half), scratch))) break;
        for (int i = 0; i < p->n_layers && !rc; i++) {
            PerLayerWeight* L = &w->layers[i];
            if ((rc = upload_qweight(&L->wq_q, file, p->dim, p->dim, scratch))) break;
            if ((rc = upload_qweight(&L->wq_k, file, p->dim, kv_dim, scratch))) break;
            if ((rc = upload_qweight(&L->wq_v, file, p->dim, kv_dim, scratch))) break;
            if ((rc = upload_qweight(&L->wq_o, file, p->dim, p->dim, scratch))) break;
            if ((rc = upload_qweight(&L->wq_up, file, p->dim, p->hidden_dim, scratch))) break;      // :191 up first
            if ((rc = upload_qweight(&L->wq_gate, file, p->dim, p->hidden_dim, scratch))) break;    // :192
            if ((rc = upload_qweight(&L->wq_down, file, p->hidden_dim, p->dim, scratch))) break;
            if ((rc = read_to_device(L->rms_att_weight, file, (size_t)p->dim * sizeof(q4_half), scratch))) break;
            if ((rc = read_to_device(L->rms_ffn_weight, file, (size_t)p->dim * sizeof(q4_half), scratch))) break;
        }
    } while (0);
    if (scratch_pinned) hipHostFree(scratch); else free(scratch);
    fclose(file);
    if (rc) { hipFree(slabs.weights); free(w->layers); memset(t, 0, sizeof(*t)); return rc; }
    if (!g_quiet) printf("done!\n");
    if ((rc = down_strip_prepare()) || (rc = cls_strip_prepare()) || (rc = ffn_pair_prepare())) { hipFree(slabs.weights); free(w->layers); memset(t, 0, sizeof(*t)); return rc; }   // (an attribute call: before any graph capture)

    // malloc_run_state :38-67. att holds n_heads*max(seq_len, dim) halves (the reference's n_heads*dim overflows
    // for seq_len > dim); this build's attention does not use it at all.
    RunState* s = &t->state;
    SlabCursor sd = {nullptr, 0};
    for (int pass = 0; pass < 2; pass++) {
        SlabCursor& c = pass == 0 ? sd : cur;
        if (pass == 1) {
            if (hipMalloc(&slabs.state, sd.off) != hipSuccess) { printf("malloc failed for allocaing run state!\n"); rc = Q4_ERR_ALLOC; break; }
            cur = {(char*)slabs.state, 0};
            // zeroed on the launch stream and waited for: a null-stream hipMemset returns before the device has finished, and nothing orders it
            // before the first launches on a non-blocking g_stream -- a small model's first K / V rows could be zeroed AFTER they were written
            if (hipMemsetAsync(slabs.state, 0, sd.off, g_stream) != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess) {
                (void)hipGetLastError();
                hipDeviceSynchronize();
                hipMemset(slabs.state, 0, sd.off);
                hipDeviceSynchronize();
            }
        }
        s->x = (q4_half*)c.take((size_t)p->dim * sizeof(q4_half));
        s->xb = (q4_half*)c.take((size_t)p->dim * sizeof(q4_half));
        s->hb = (q4_half*)c.take((size_t)p->hidden_dim * sizeof(q4_half));
        s->q = (q4_half*)c.take((size_t)p->dim * sizeof(q4_half));
        s->att = (q4_half*)c.take(att_buffer_bytes(p));
        s->logits = (q4_half*)c.take((size_t)p->vocab_size * sizeof(q4_half));
        s->key_cache = (q4_half*)c.take(sizeof(q4_half) * p->n_layers * p->seq_len * kv_dim);
        s->value_cache = (q4_half*)c.take(sizeof(q4_half) * p->n_layers * p->seq_len * kv_dim);
        s->pos = (int*)c.take(sizeof(int));
    }
    if (!rc && hipHostMalloc(&slabs.shared, sizeof(SharedData), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {   // cudaMallocHost :50
        printf("malloc failed for allocaing run state!\n");
        rc = Q4_ERR_ALLOC;
    }
    if (!rc) {
        s->shared_data = (SharedData*)slabs.shared;
        s->shared_data->pos = 0;
        memset((void*)s->shared_data->tokens, 0, sizeof(int) * 16);
    }
    if (!rc && perplexity) {                                                       // :60-66
        if (hipMalloc(&slabs.logits_array, sizeof(float) * p->seq_len * p->vocab_size) != hipSuccess) {
            printf("malloc failed for allocaing logits_array!\n");
            rc = Q4_ERR_ALLOC;
        }
        s->logits_array = (float*)slabs.logits_array;
    }
    if (rc) {
        if (slabs.weights) hipFree(slabs.weights);
        if (slabs.state) hipFree(slabs.state);
        if (slabs.shared) hipHostFree(slabs.shared);
        free(w->layers);
        memset(t, 0, sizeof(*t));
        return rc;
    }
    // fused QKV + RoPE epilogue (multi-head and grouped-query models alike): table of this model's own seq_len
    rc = rope_table_build(&slabs.rope_table, p->seq_len, p->dim / p->n_heads, p->rope_theta);
    g_slabs[t] = slabs;
    if (rc) { q4_free_transformer(t); return rc; }
    if (slabs.rope_table) g_rope_by_state[&t->state] = slabs.rope_table;
    const size_t sync_words = ffn_pair_sync_offset(p->dim) + ffn_pair_sync_words(p->dim, p->hidden_dim);
    if (hipMalloc((void**)&slabs.sync, sync_words * sizeof(unsigned)) == hipSuccess) {
        // zeroed on the launch stream and waited for: nothing else orders a null-stream memset before the first launch on a
        // non-blocking g_stream when the caller starts with q4_run_transformer instead of q4_reset_sequence
        if (hipMemsetAsync(slabs.sync, 0, sync_words * sizeof(unsigned), g_stream) != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess) {
            (void)hipGetLastError();
            hipDeviceSynchronize();
            hipMemset(slabs.sync, 0, sync_words * sizeof(unsigned));
            hipDeviceSynchronize();
        }
        g_slabs[t] = slabs;
        g_sync_by_state[&t->state] = slabs.sync;
        g_sync_words[&t->state] = sync_words;
        g_att_bytes[&t->state] = att_buffer_bytes(p);
        measure_kv_price(p, &t->state, slabs.sync);
    } else {
        (void)hipGetLastError();    // no hand-off words: the network runs its five-launch sequence
    }
    return Q4_OK;
}

void q4_free_transformer(Transformer* t) {                                        // :428-432
    if (!t) return;
    for (GraphSet& gs : g_sets)
        if (gs.owner == (const void*)&t->state) drop_graphs(gs, false);   // the owner is recorded as the RunState (run_transformer_at)
    auto it = g_slabs.find(t);
    if (it != g_slabs.end()) {
        hipDeviceSynchronize();
        if (it->second.weights) hipFree(it->second.weights);
        if (it->second.state) hipFree(it->second.state);
        if (it->second.shared) hipHostFree(it->second.shared);
        if (it->second.logits_array) hipFree(it->second.logits_array);
        if (it->second.rope_table) hipFree(it->second.rope_table);
        if (it->second.sync) { attention_set_kv_price(it->second.sync, 0.0); hipFree(it->second.sync); }
        g_rope_by_state.erase(&t->state);
        g_sync_by_state.erase(&t->state);
        g_sync_words.erase(&t->state);
        g_att_bytes.erase(&t->state);
        g_slabs.erase(it);
    }
    free(t->weights.layers);
    memset(t, 0, sizeof(*t));
}

double q4_kv_stream_price(const RunState* s) { unsigned* sync = sync_words_of(s); return sync ? attention_get_kv_price(sync) : 0.0; }

Transformer* q4_transformer_new(const char* checkpoint_path, int perplexity, int* status) {
    Transformer* t = (Transformer*)calloc(1, sizeof(Transformer));
    int rc = q4_build_transformer(t, checkpoint_path, perplexity);
    if (status) *status = rc;
    if (rc) { free(t); return nullptr; }
    // q4_build_transformer registered the slabs under `t` already
    return t;
}
void q4_transformer_delete(Transformer* t) {
    if (!t) return;
    q4_free_transformer(t);
    free(t);
}
const Config* q4_transformer_config(const Transformer* t) { return &t->config; }
RunState* q4_transformer_state(Transformer* t) { return &t->state; }
TransformerWeights* q4_transformer_weights(Transformer* t) { return &t->weights; }

// ---------------------------------------------------------------------------------------------------
// run_llama_network, llama2_q4.cu:286-340
#define Q4_TRY(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

// measurement knob (tools/lab/breakdown.py): leave out a class of launches to read its marginal cost inside the token
// graph. Results are garbage with any bit set; never set by the product path.
#ifdef Q4_PROFILING
static int g_skip = 0;
void q4_set_skip_mask(int mask) { g_skip = mask; q4_reset_graphs(); }
#else
enum { g_skip = 0 };   // the shipped library cannot leave launches out
#endif
// in-network timing (q4_bench_in_network): launches of the class whose bit is in g_time_mask carry dispatch timestamps
static int g_time_mask = 0;
static std::vector<hipEvent_t>* g_time_events = nullptr;
static std::vector<int>* g_time_bits = nullptr;
static inline void arm_timing(int bit) {
    g_ev_start = g_ev_stop = nullptr;
    if (!(g_time_mask & bit) || !g_time_events) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_time_events->push_back(a);
    g_time_events->push_back(b);
    if (g_time_bits) g_time_bits->push_back(bit);
    g_ev_start = a;
    g_ev_stop = b;
}
#define Q4_UNLESS(bit, call) do { if (!(g_skip & (bit))) { if (g_time_mask) arm_timing(bit); int rc__ = (call); g_ev_start = g_ev_stop = nullptr; if (rc__) return rc__; } } while (0)

static int run_network(const int* pPos, const Config* p, RunState* s, const TransformerWeights* w, int seq_len_bin, bool have_embedding);
#ifdef Q4_PROFILING
// tools/error_growth.py: the residual stream after the attention half and after the FFN half of every layer, [n_layers][2][dim] halves on
// the device (eager launches only: a copy node would be baked into captured graphs)
static q4_half* g_layer_dump = nullptr;
extern "C" void q4_set_layer_dump(void* p) { g_layer_dump = (q4_half*)p; }
#define Q4_LAYER_DUMP(half_) do { if (g_layer_dump) Q4_HIP(hipMemcpyAsync(g_layer_dump + (size_t)(2 * l + (half_)) * dim, x, (size_t)dim * sizeof(q4_half), hipMemcpyDeviceToDevice, g_stream)); } while (0)
#else
#define Q4_LAYER_DUMP(half_) do { } while (0)
#endif
int q4_run_llama_network(const int* pPos, const Config* p, RunState* s, const TransformerWeights* w, int seq_len_bin) {
    return run_network(pPos, p, s, w, seq_len_bin, false);
}
// have_embedding: the preceding launch of the stream (the greedy sampler of the previous step, inside one graph replay) has left
// the token's embedding row in s->x already
static int run_network(const int* pPos, const Config* p, RunState* s, const TransformerWeights* w, int seq_len_bin, bool have_embedding) {
    q4_half* x = s->x;
    const int dim = p->dim;
    const int hidden_dim = p->hidden_dim;
    const int head_size = dim / p->n_heads;
    const int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    const int kv_mul = p->n_heads / p->n_kv_heads;
    const float2* rope_table = rope_table_of(s);
    unsigned* sync = sync_words_of(s);

    if (!have_embedding)
        Q4_UNLESS(64, q4_copy_embedding(x, w->token_embedding_table, dim, s->shared_data->tokens, pPos));   // :294

    const size_t att_bytes = att_buffer_bytes(p);
    // :320-323 as ONE launch where the geometry, the bin and the stream's CUs admit it (layer_attn.h); the launch in front of it
    // (the fused QKV GEMV) advances its epoch word
    const bool ao = g_fusion >= 3 && sync &&
                    attention_oproj_form(dim, kv_dim, head_size, p->n_heads, seq_len_bin, s->att != nullptr, att_bytes, g_att_split_min, g_att_chunk) >= 0;

    // :326-332 as ONE launch (gemv_ffn_pair.h) where the shapes and the stream admit it; its tag is the same epoch word
    const bool fp = g_fusion >= 4 && sync && ffn_pair_covers(dim, hidden_dim);
    // ... and the next layer's :300-317 with it (fusion level 5): that layer then has no QKV launch of its own
    const bool fq = fp && g_fusion >= 5 && ffn_qkv_covers(dim, hidden_dim, kv_dim, head_size, rope_table != nullptr);
    // ... and THIS layer's :320-323 in front of it (fusion level 6): the whole layer behind its q / k / v is one launch. Only where level 3 would run the
    // V-slice role (forms 5 / 6, bins <= 256), whose arithmetic the launch's first phase repeats
    const int ao_form = fq ? attention_oproj_form(dim, kv_dim, head_size, p->n_heads, seq_len_bin, s->att != nullptr, att_bytes, g_att_split_min, g_att_chunk) : -1;
    const bool fa = fq && g_fusion >= 6 && (ao_form == 5 || ao_form == 6) && layer_att_covers(dim, hidden_dim, kv_dim, p->n_heads, seq_len_bin) && !(g_skip & 15);
    bool qkv_done = false;                             // the previous launch has left q and the K / V rows of this layer

    for (int l = 0; l < p->n_layers; l++) {
        const PerLayerWeight* L = &w->layers[l];
        // :303. 64-bit: the reference's int overflows at e.g. 13B x 16384 positions (40 * 16384 * 5120 > 2^31); the
        // int-typed entry points of the 1:1 path get pre-offset cache pointers and loff = 0 instead
        const long long loff = (long long)l * p->seq_len * kv_dim;
        if (qkv_done) {
            qkv_done = false;
        } else if (g_fusion) {
            // rmsnorm (:300) + qkv (:307, or the three GEMVs of the GQA branch :310-312) + RoPE (:317) in one launch
            Q4_UNLESS(1, launch_qkv_fused(s->q, s->key_cache, s->value_cache, x, L->rms_att_weight, &L->wq_q, &L->wq_k, &L->wq_v,
                                          dim, kv_dim, loff, pPos, head_size, p->rope_theta, rope_table, (ao || fp) ? sync + SYNC_EPOCH : nullptr));
        } else {
            Q4_TRY(q4_rmsnorm(s->xb, x, L->rms_att_weight, dim));                                      // :300
            if (dim == kv_dim) {
                Q4_TRY(q4_qkv_matvec(s->q, s->key_cache + loff, s->value_cache + loff, s->xb, &L->wq_q, &L->wq_k, &L->wq_v, dim, dim, 0, pPos));
            } else {
                Q4_TRY(q4_matmul_q4(s->q, s->xb, &L->wq_q, dim, dim, 0, -1, nullptr));                 // :310-312
                Q4_TRY(q4_matmul_q4(s->key_cache + loff, s->xb, &L->wq_k, dim, kv_dim, 0, 0, pPos));
                Q4_TRY(q4_matmul_q4(s->value_cache + loff, s->xb, &L->wq_v, dim, kv_dim, 0, 0, pPos));
            }
            Q4_TRY(q4_rope_rotation(s->q, s->key_cache + loff, p->n_heads, p->n_kv_heads, head_size, pPos, 0, p->rope_theta));   // :317
        }
        if (fa) {
            // (phases A and O of the launch below)
        } else if (ao) {
            Q4_UNLESS(6, launch_attention_oproj(x, s->xb, s->q, s->key_cache + loff, s->value_cache + loff, &L->wq_o, dim, kv_dim, p->n_heads,
                                                pPos, seq_len_bin, sync, (float*)s->att, att_bytes, g_att_split_min, g_att_chunk));
        } else {
        Q4_UNLESS(2, launch_attention(s->xb, s->q, s->key_cache + loff, s->value_cache + loff, p->n_heads, head_size, kv_mul,
                                      seq_len_bin, pPos, (float*)s->att,
                                      att_bytes, sync && p->n_heads <= SYNC_MAX_HEADS ? sync + SYNC_ARRIVE : nullptr));   // :320
        Q4_UNLESS(4, q4_matmul_q4(s->x, s->xb, &L->wq_o, dim, dim, 1, -1, nullptr));                   // :323
        }
        Q4_LAYER_DUMP(0);
        if (fp) {
            FfnQkvNext nx = {};
            const bool with_next = fq && l + 1 < p->n_layers && !(g_skip & 9);
            if (with_next) {
                const PerLayerWeight* N = &w->layers[l + 1];
                const long long noff = (long long)(l + 1) * p->seq_len * kv_dim;
                nx = FfnQkvNext{N->rms_att_weight, &N->wq_q, &N->wq_k, &N->wq_v, s->q, s->key_cache + noff, s->value_cache + noff, pPos, rope_table,
                                sync + SYNC_EPOCH, kv_dim, head_size};
            }
            const FfnLayerAtt la = {s->xb, s->q, s->key_cache + loff, s->value_cache + loff, &L->wq_o, pPos, p->n_heads, kv_dim, seq_len_bin};
            Q4_UNLESS(8, launch_ffn_pair(x, s->hb, L->rms_ffn_weight, &L->wq_gate, &L->wq_up, &L->wq_down, dim, hidden_dim, sync, ffn_pair_sync_offset(dim), 0u,
                                         with_next ? &nx : nullptr, fa ? &la : nullptr));   // :326-332 (+ the next layer's :300-317, + this layer's :320-323)
            qkv_done = with_next;
            Q4_LAYER_DUMP(1);
            continue;
        }
        if (g_fusion) {
            Q4_UNLESS(8, launch_ffn_fused(s->hb, x, L->rms_ffn_weight, &L->wq_gate, &L->wq_up, dim, hidden_dim));   // :326 + :329
        } else {
            Q4_TRY(q4_rmsnorm(s->xb, x, L->rms_ffn_weight, dim));                                      // :326
            Q4_TRY(q4_ffn_matvec_silu(s->hb, s->xb, &L->wq_gate, &L->wq_up, dim, hidden_dim));         // :329
        }
        Q4_UNLESS(16, q4_matmul_q4(s->x, s->hb, &L->wq_down, hidden_dim, dim, 1, -1, nullptr));        // :332
        Q4_LAYER_DUMP(1);
    }
    if (g_fusion >= 1) {      // one launch where the classifier runs as strips (gemv_strip_cls.h): the final norm inside its x staging
        Q4_UNLESS(32, classifier_with_final_norm(s->logits, x, w->rms_final_weight, w->wcls, p->dim, p->vocab_size));
    } else {
        Q4_UNLESS(32, q4_rmsnorm(x, x, w->rms_final_weight, dim));                                         // :336
        Q4_UNLESS(32, q4_matmul_f16(s->logits, x, w->wcls, p->dim, p->vocab_size, 1, 0, 0, 0, -1, 1.0f));  // :339
    }
    return Q4_OK;
}

// Average duration of one launch class INSIDE the eager decode network (x produced by the previous kernel, caches
// in the state the real token loop leaves them): `tokens` decode steps from the current position with dispatch
// timestamps on the launches of time_mask | report_mask (1 qkv, 2 attention, 4 o-proj, 8 gate/up, 16 down, 32 final
// norm + classifier, 64 embedding); the statistics cover report_mask. The launches are the product's own, only
// hipExtLaunchKernelGGL carries the events.
double q4_bench_in_network(int time_mask, int report_mask, const Config* p, RunState* s, const TransformerWeights* w,
                           int tokens, double* min_us, double* max_us, int* launches) {
    if (!p || !s || !w || tokens < 1 || !g_stream) return -1.0;
    std::vector<hipEvent_t> ev;
    std::vector<int> bits;
    int rc = 0;
    const int pos0 = s->shared_data->pos;
    for (int t = 0; t < tokens && !rc; t++) {
        const int pos = pos0 + t;
        if (pos + 1 >= p->seq_len) break;
        g_time_events = &ev;
        g_time_bits = &bits;
        g_time_mask = time_mask | report_mask;
        rc = q4_run_llama_network(s->pos, p, s, w, pos + 1);
        g_time_mask = 0;
        g_time_events = nullptr;
        g_time_bits = nullptr;
        if (!rc) rc = q4_argmax(s->logits, p->vocab_size, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos, 1);
    }
    if (!rc && hipStreamSynchronize(g_stream) != hipSuccess) rc = Q4_ERR_HIP;
    double total = 0, mn = 1e30, mx = 0;
    int n = 0;
    for (size_t i = 0; i < bits.size() && !rc; i++) {
        if (!(bits[i] & report_mask)) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { rc = Q4_ERR_HIP; break; }
        const double us = ms * 1000.0;
        total += us;
        n++;
        if (us < mn) mn = us;
        if (us > mx) mx = us;
    }
    for (auto& e : ev) hipEventDestroy(e);
    if (rc || n == 0) return -1.0;
    if (min_us) *min_us = mn;
    if (max_us) *max_us = mx;
    if (launches) *launches = n;
    return total / n;
}

// ---------------------------------------------------------------------------------------------------
// sampler.h
int build_sampler(Sampler* sampler, int vocab_size, float temperature, float topp, unsigned long long rng_seed) {
    memset(sampler, 0, sizeof(*sampler));
    sampler->vocab_size = vocab_size;
    sampler->temperature = temperature;
    sampler->topp = topp;
    sampler->rng_state = rng_seed;
    Q4_HIP(hipMalloc((void**)&sampler->indices, vocab_size * sizeof(int)));        // sampler.h:22
    return Q4_OK;
}
// Coins of the sampled steps that run inside captured graphs: the xorshift stream stays on the host (sampler.h:31-40), the values
// of the steps a replay covers are copied to a device ring indexed by position just before the replay (one small async copy per
// replay), and topp_sample_kernel reads coins[position] -- so the launch needs no per-step argument and sampled steps go out
// eight per replay like greedy ones. (The Sampler struct is the reference's: the ring lives beside it.)
struct CoinRing { float* host; float* dev; int cap; };
static std::map<const Sampler*, CoinRing> g_coin_rings;
static int coin_ring_for(const Sampler* sampler, int positions, CoinRing** out) {
    CoinRing& r = g_coin_rings[sampler];
    if (r.cap < positions) {
        if (r.host) hipHostFree(r.host);
        if (r.dev) hipFree(r.dev);
        r = CoinRing{nullptr, nullptr, 0};
        Q4_HIP(hipHostMalloc((void**)&r.host, (size_t)positions * sizeof(float), hipHostMallocDefault));
        Q4_HIP(hipMalloc((void**)&r.dev, (size_t)positions * sizeof(float)));
        // zeroed on both sides: a caller whose `pos` runs behind the device position (q4_run_transformer_at / _steps with a stale `pos`: the
        // sampled step reads coins[device position], the host fills ring[pos + i]) then samples with coin 0.0 -- the most probable token,
        // deterministic -- instead of with uninitialised memory. INTEGRATION.md: `pos` must equal the device position for sampled steps.
        memset(r.host, 0, (size_t)positions * sizeof(float));
        Q4_HIP(hipMemsetAsync(r.dev, 0, (size_t)positions * sizeof(float), g_stream));
        r.cap = positions;
    }
    *out = &r;
    return Q4_OK;
}
void destroy_sampler(Sampler* sampler) {
    auto cr = g_coin_rings.find(sampler);
    if (cr != g_coin_rings.end()) {
        for (GraphSet& gs : g_sets)
            if (gs.sampler == sampler) drop_graphs(gs, true);
        if (g_stream) hipStreamSynchronize(g_stream);
        if (cr->second.host) hipHostFree(cr->second.host);
        if (cr->second.dev) hipFree(cr->second.dev);
        g_coin_rings.erase(cr);
    }
    if (sampler->indices) hipFree(sampler->indices);
    if (sampler->tempStorage_sort) hipFree(sampler->tempStorage_sort);
    if (sampler->tempStorage_scan) hipFree(sampler->tempStorage_scan);
    memset(sampler, 0, sizeof(*sampler));
}
unsigned int random_u32(unsigned long long* state) {                               // sampler.h:31-37
    *state ^= *state >> 12;
    *state ^= *state << 25;
    *state ^= *state >> 27;
    return (unsigned int)((*state * 0x2545F4914F6CDD1Dull) >> 32);
}
float random_f32(unsigned long long* state) { return (random_u32(state) >> 8) / 16777216.0f; }   // :38-40

Sampler* q4_sampler_new(int vocab_size, float temperature, float topp, unsigned long long rng_seed) {
    Sampler* s = (Sampler*)calloc(1, sizeof(Sampler));
    if (build_sampler(s, vocab_size, temperature, topp, rng_seed)) { free(s); return nullptr; }
    return s;
}
void q4_sampler_delete(Sampler* s) {
    if (!s) return;
    destroy_sampler(s);
    free(s);
}

__attribute__((visibility("hidden"))) int q4_sample_topp_device(Sampler* sampler, RunState* s, float coin, const float* coins, q4_half* x_next,
                                                                const q4_half* table, int dim);   // q4_sampling.hip
__attribute__((visibility("hidden"))) int q4_sample_topp_prepare(Sampler* sampler);

static bool sampler_is_greedy(const Sampler* sampler, int gen_token) {
    return sampler->temperature == 0.0f || !gen_token;                             // sampler.h:47
}

// sample(), sampler.h:43-82. `launch_argmax` false when the captured graph already contains it.
static int sample_impl(Sampler* sampler, RunState* s, int gen_token, bool launch_argmax) {
    float coin = random_f32(&sampler->rng_state);                                  // :45, drawn every step (P8)
    if (sampler_is_greedy(sampler, gen_token)) {
        if (launch_argmax)
            return q4_argmax(s->logits, sampler->vocab_size, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos, gen_token);
        return Q4_OK;
    }
    return q4_sample_topp_device(sampler, s, coin, nullptr, nullptr, nullptr, 0);
}
int q4_sample(Sampler* sampler, RunState* s, int gen_token) { return sample_impl(sampler, s, gen_token, true); }

// ---------------------------------------------------------------------------------------------------
// run_transformer, llama2_q4.cu:346-395
int q4_run_transformer(int gen_token, const Config* p, RunState* s, const TransformerWeights* w, int copyLogits,
                       Sampler* pSampler) {
    return q4_run_transformer_at(s->shared_data->pos, gen_token, p, s, w, copyLogits, pSampler);   // :354
}

// The same step with the position supplied by the caller instead of read back from SharedData::pos: the device keeps
// its own position (`s->pos`) and reads its input token from the ring, so step pos+1 can be queued while step pos is
// still running (generate() below); only the graph bin depends on the host's idea of the position.
static int graph_bin(int seq_len, const Config* p, int* seq_len_bin_out) {
    int graphIndex;
    int seq_len_bin = 128;
    for (graphIndex = 0; graphIndex < Q4_MAX_GRAPHS - 1; seq_len_bin *= 2, graphIndex++)
        if (seq_len <= seq_len_bin) break;                                         // :356-359
    if ((seq_len > seq_len_bin) || (graphIndex == Q4_MAX_GRAPHS - 1)) seq_len_bin = p->seq_len;   // :360
    *seq_len_bin_out = seq_len_bin;
    return graphIndex;
}

int q4_run_transformer_at(int pos, int gen_token, const Config* p, RunState* s, const TransformerWeights* w, int copyLogits,
                          Sampler* pSampler) {
    return q4_run_transformer_steps(pos, 1, gen_token, p, s, w, copyLogits, pSampler);
}

// `nsteps` consecutive greedy steps (positions pos .. pos + nsteps - 1, same gen_token) as ONE graph replay: the device
// advances its own position and feeds itself the tokens (argmax_kernel writes the ring, copy_embedding reads it), so a
// replay needs nothing from the host between steps -- the per-replay launch cost is paid once per nsteps tokens. Only
// nsteps == 1 or Q4_MULTI_STEPS are captured; the caller keeps a group inside one sequence-length bin (q4_steps_that_fit).
int q4_run_transformer_steps(int pos, int nsteps, int gen_token, const Config* p, RunState* s, const TransformerWeights* w,
                             int copyLogits, Sampler* pSampler) {
    const int seq_len = pos + nsteps;                                              // :354 (of the group's last step)
    const bool greedy = sampler_is_greedy(pSampler, gen_token);
    int seq_len_bin;
    const int graphIndex = graph_bin(seq_len, p, &seq_len_bin);
    if (nsteps != 1 && (nsteps != g_multi_steps || g_use_graphs != 1)) return Q4_ERR_ARG;
    if (pos < 0 || pos + nsteps > p->seq_len) return Q4_ERR_ARG;

    if (g_use_graphs == 1) {
        GraphSet& gs = graph_set_of(s);
        // Unlike the reference, the greedy sampler kernel and the fp32 logits copy are part of the captured
        // graph (one launch per token instead of up to three); the variant index keeps them apart.
        const int variant = (gen_token ? 1 : 0) | (copyLogits ? 2 : 0) | (greedy ? 0 : 4) | (nsteps > 1 ? 8 : 0);
        CoinRing* ring = nullptr;
        if (!greedy) {     // the sampling kernel is part of the graph: its temperature, top-p, scratch and coin ring are baked in
            Q4_TRY(coin_ring_for(pSampler, p->seq_len, &ring));
            Q4_TRY(q4_sample_topp_prepare(pSampler));     // scratch + LDS opt-in: not capturable
            if (gs.sampler != pSampler || gs.temperature != pSampler->temperature || gs.topp != pSampler->topp || gs.coins != ring->dev) {
                drop_graphs(gs, true);
                gs.sampler = pSampler; gs.temperature = pSampler->temperature; gs.topp = pSampler->topp; gs.coins = ring->dev;
            }
        }
        if (!gs.captured[graphIndex][variant]) {                                   // :362-371
            hipGraph_t graph = nullptr;
            Q4_HIP(hipStreamBeginCapture(g_stream, hipStreamCaptureModeGlobal));
            int rc = 0;
            // generated tokens, several steps per replay: step i + 1 takes the token step i writes -- its sampler launch leaves the
            // embedding row in s->x and the copy_embedding launch of step i + 1 is left out (the fused sequence only: level 0 is
            // the reference's 1:1 launch list)
            const bool feed = gen_token && nsteps > 1 && g_fusion >= 1;
            for (int i = 0; i < nsteps && !rc; i++) {
                rc = run_network(s->pos, p, s, w, seq_len_bin, feed && i > 0);
                if (!rc && copyLogits) rc = q4_copy_logits_at_pos(s->logits_array, s->logits, p->vocab_size, s->pos);
                if (!rc && greedy) {
                    if (feed && i + 1 < nsteps)
                        rc = launch_argmax_feed(s->logits, p->vocab_size, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos,
                                                s->x, w->token_embedding_table, p->dim);
                    else
                        rc = q4_argmax(s->logits, p->vocab_size, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos, gen_token);
                } else if (!rc) {   // sampler.h:51-81 inside the graph: the coin comes from the ring, by position
                    rc = q4_sample_topp_device(pSampler, s, 0.f, ring->dev, feed && i + 1 < nsteps ? s->x : nullptr, w->token_embedding_table, p->dim);
                }
            }
            hipError_t e = hipStreamEndCapture(g_stream, &graph);
            if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
            Q4_HIP(e);
            Q4_HIP(hipGraphInstantiate(&gs.exec[graphIndex][variant], graph, nullptr, nullptr, 0));
            Q4_HIP(hipGraphDestroy(graph));
            gs.captured[graphIndex][variant] = true;
            g_graph_captures++;
        }
        // one coin per step, drawn whether the step samples or not (sampler.h:45, P8); the sampled steps' coins go to the ring
        for (int i = 0; i < nsteps; i++) {
            const float coin = random_f32(&pSampler->rng_state);
            if (ring) ring->host[pos + i] = coin;
        }
        if (ring) Q4_HIP(hipMemcpyAsync(ring->dev + pos, ring->host + pos, (size_t)nsteps * sizeof(float), hipMemcpyHostToDevice, g_stream));
        Q4_HIP(hipGraphLaunch(gs.exec[graphIndex][variant], g_stream));          // :372 (:384: the sampler launch is in the graph)
        return Q4_OK;
    }
    Q4_TRY(run_network(s->pos, p, s, w, g_use_graphs == 2 ? seq_len_bin : seq_len, false));   // :374
    if (copyLogits) Q4_TRY(q4_copy_logits_at_pos(s->logits_array, s->logits, p->vocab_size, s->pos));   // :377-382
    return sample_impl(pSampler, s, gen_token, true);
}

// ---------------------------------------------------------------------------------------------------
int q4_reset_sequence(RunState* s, const int* prompt_tokens, int num_prompt_tokens) {
    Q4_HIP(hipMemsetAsync(s->pos, 0, sizeof(int), g_stream));                     // llama2_q4.cu:461
    if (g_rearm_after > 0 && --g_rearm_after == 0 && g_fusion == 1) { g_fusion = g_rearm_level; q4_reset_graphs(); }   // probation over
    Q4_TRY(clear_handoff_state(s, false));     // counters and granules; the error word [0] stays until q4_handoff_status reads it
    Q4_HIP(hipStreamSynchronize(g_stream));
    s->shared_data->pos = 0;                                                       // :462
    if (prompt_tokens && num_prompt_tokens > 0)
        memcpy((void*)s->shared_data->tokens, prompt_tokens, sizeof(int) * num_prompt_tokens);   // :463
    return Q4_OK;
}
int q4_shared_pos(const RunState* s) { return s->shared_data->pos; }

// How many steps a token loop may queue at once from `pos`: Q4_MULTI_STEPS when graphs are on, the sampler is greedy, the
// whole group generates (or the whole group feeds prompt tokens), ends by `steps` and stays inside one sequence-length
// bin; else 1.
int q4_steps_that_fit(int pos, int num_prompt_tokens, int steps, const Config* p, const Sampler* sampler) {
    const int k = g_multi_steps;
    if (k <= 1 || g_use_graphs != 1 || pos + k > steps || pos + k > p->seq_len) return 1;
    const bool gen0 = pos >= num_prompt_tokens - 1, gen1 = pos + k - 1 >= num_prompt_tokens - 1;
    if (gen0 != gen1) return 1;
    (void)sampler;     // sampled steps take their coins from a device ring by position: they go out k per replay too
    int b0, b1;
    if (graph_bin(pos + 1, p, &b0) != graph_bin(pos + k, p, &b1)) return 1;
    return k;
}
// The in-launch hand-offs of fusion levels 3 and 4 (attention -> o-proj, layer_attn.h; the FFN pair launch, gemv_ffn_pair.h) spin for a bounded time; a spin that ran out
// sets the model's error word: everything computed since is invalid. Synchronises the stream and reports it ONCE: the word,
// the counters and the granules are cleared (the epoch keeps counting: tags never repeat), and the library drops to fusion
// level 1 (no in-launch waits), so the caller can simply redo the sequence -- the token loops of this library do exactly that.
// The level in force comes back by itself after 16 sequences (q4_reset_sequence counts them; 32, 64, ... after further time-outs) or at
// once with q4_set_fusion(level).
int q4_handoff_status(const RunState* s) {
    Q4_HIP(hipStreamSynchronize(g_stream));
    unsigned* sync = sync_words_of(s);
    auto it = g_sync_words.find(s);
    if (!sync || it == g_sync_words.end()) return Q4_OK;
    unsigned flag = 0;
    Q4_HIP(hipMemcpy(&flag, sync + SYNC_ERROR, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
        Q4_TRY(clear_handoff_state(s, true));
        g_handoff_timeouts++;
        if (g_fusion >= 3) {    // one transient stall (a profiler attaching, a co-tenant) must not cost every later sequence its 3 %
            g_rearm_level = g_fusion;
            g_fusion = 1;
            g_rearm_after = g_rearm_backoff;
            if (g_rearm_backoff < (1 << 20)) g_rearm_backoff *= 2;
            q4_reset_graphs();
        }
        snprintf(g_last_error, sizeof(g_last_error), "an in-launch hand-off timed out (fusion level %d); state cleared, continuing at fusion level 1 for the next %d sequences", g_rearm_level, g_rearm_after);
        if (!g_quiet) fprintf(stderr, "llama2_q4: %s\n", g_last_error);
        return Q4_ERR_HIP;
    }
    return Q4_OK;
}
int q4_handoff_timeouts(void) { return g_handoff_timeouts; }
// Wait until the device has published position >= pos (argmax_kernel / sample_scan_kernel write the token, fence, then
// SharedData::pos -- "unblocks the CPU", gpu_kernels.h:490). Spins on the pinned word; falls back to the stream state
// so that a failed launch cannot hang the host.
int q4_wait_pos(const RunState* s, int pos) {
    volatile int* p = &s->shared_data->pos;
    for (unsigned spins = 1;; spins++) {
        if (*p >= pos) return Q4_OK;
        if ((spins & 0x3fff) == 0) {
            hipError_t e = hipStreamQuery(g_stream);
            if (e == hipSuccess) return *p >= pos ? Q4_OK : Q4_ERR_ARG;           // stream drained: pos is final
            if (e != hipErrorNotReady) Q4_HIP(e);
        }
        __builtin_ia32_pause();
    }
}
int q4_shared_token(const RunState* s, int index) { return s->shared_data->tokens[index]; }

int q4_get_logits(const Transformer* t, q4_half* host_out) {
    Q4_HIP(hipStreamSynchronize(g_stream));
    Q4_HIP(hipMemcpy(host_out, t->state.logits, (size_t)t->config.vocab_size * sizeof(q4_half), hipMemcpyDeviceToHost));
    return Q4_OK;
}
int q4_get_kv_row(const Transformer* t, int layer, int pos, q4_half* host_k, q4_half* host_v) {
    const Config* p = &t->config;
    const int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    const size_t off = (size_t)layer * p->seq_len * kv_dim + (size_t)pos * kv_dim;
    Q4_HIP(hipStreamSynchronize(g_stream));
    Q4_HIP(hipMemcpy(host_k, t->state.key_cache + off, (size_t)kv_dim * sizeof(q4_half), hipMemcpyDeviceToHost));
    Q4_HIP(hipMemcpy(host_v, t->state.value_cache + off, (size_t)kv_dim * sizeof(q4_half), hipMemcpyDeviceToHost));
    return Q4_OK;
}
int q4_get_logits_array(const Transformer* t, int num_pos, float* host_out) {
    if (!t->state.logits_array) return Q4_ERR_ARG;
    Q4_HIP(hipStreamSynchronize(g_stream));
    Q4_HIP(hipMemcpy(host_out, t->state.logits_array, (size_t)num_pos * t->config.vocab_size * sizeof(float), hipMemcpyDeviceToHost));
    return Q4_OK;
}


// generate() llama2_q4.cu:436-492 on token ids (no tokenizer, no printing): same loop order -- synchronise,
// launch step `pos`, then look at the token produced by the PREVIOUS step; same throughput rule (pos-1)/elapsed.
double q4_generate_ids(Transformer* t, Sampler* sampler, const int* prompt_tokens, int num_prompt_tokens, int steps,
                       int* out_tokens, int* timed_tokens_out, double* seconds_out) {
    if (num_prompt_tokens < 1) return -1.0;
    if (steps <= 0 || steps > t->config.seq_len) steps = t->config.seq_len;        // :690
    const unsigned long long rng0 = sampler->rng_state;
    for (int attempt = 0;; attempt++) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        int pos = 0, queued = 0, group_start = 0;
        unsigned long long group_rng = sampler->rng_state;   // sampler state in front of the group of steps queued last
        if (q4_reset_sequence(&t->state, prompt_tokens, num_prompt_tokens)) return -1.0;
        bool stopped = false;
        while (pos < steps) {
            // the reference synchronises and then launches step `pos` (:468-470); here the launch goes out first, queued
            // behind step pos-1, and the host then waits for step pos-1's token -- same device order, no idle gap per token.
            // Greedy steps inside one bin go out Q4_MULTI_STEPS at a time (one graph replay); a stop at EOS leaves at most
            // Q4_MULTI_STEPS - 1 surplus steps behind, which the next q4_reset_sequence discards.
            if (pos >= queued) {
                const int k = q4_steps_that_fit(pos, num_prompt_tokens, steps, &t->config, sampler);
                group_start = pos;
                group_rng = sampler->rng_state;
                if (q4_run_transformer_steps(pos, k, pos >= num_prompt_tokens - 1, &t->config, &t->state, &t->weights, 0, sampler)) return -1.0;
                queued = pos + k;
            }
            if (q4_wait_pos(&t->state, pos)) return -1.0;                              // :468
            if (pos > 0) {
                int next = t->state.shared_data->tokens[pos];                          // :473
                if (next >= t->config.vocab_size) next = 0;                            // :474
                if (next == 2) { stopped = true; break; }                              // eos_token, :477
            }
            pos++;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);                                           // :485, taken where the reference takes it
        if (stopped) {
            // the reference has drawn one coin per run_transformer call, steps 0..pos (sampler.h:45); a multi-step group drew
            // for its surplus steps too: put the stream back to exactly pos + 1 draws so a reused Sampler continues like the reference's
            sampler->rng_state = group_rng;
            for (int i = group_start; i <= pos; i++) (void)random_f32(&sampler->rng_state);
        }
        const double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        const int timed_tokens = pos - 1;                                              // :488
        if (q4_handoff_status(&t->state)) {     // a timed-out in-launch wait: the library is at fusion level 1 now, state cleared
            if (attempt == 0) { sampler->rng_state = rng0; continue; }                 // redo the whole sequence once
            return -1.0;
        }
        if (out_tokens) {
            const int n = (pos < steps ? pos : steps) + 1;
            for (int i = 0; i < n && i < Q4_MAX_SEQ_LEN; i++) out_tokens[i] = t->state.shared_data->tokens[i];
        }
        if (timed_tokens_out) *timed_tokens_out = timed_tokens;
        if (seconds_out) *seconds_out = secs;
        return secs > 0 ? timed_tokens / secs : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------
// perplexity.h:3-51 (host math)
void q4_softmax_f32(float* x, int size) {
    float max_val = x[0];
    for (int i = 1; i < size; i++)
        if (x[i] > max_val) max_val = x[i];
    float sum = 0.0f;
    for (int i = 0; i < size; i++) {
        x[i] = expf(x[i] - max_val);
        sum += x[i];
    }
    for (int i = 0; i < size; i++) x[i] /= sum;
}
float compute_perplexity(const int* tokens, float* logits, int num_tokens, int vocab_size) {
    double sum = 0.0;
    for (int i = 0; i < num_tokens; i++) {
        int word_index = tokens[i];
        q4_softmax_f32(&logits[(size_t)i * vocab_size], vocab_size);
        double prob = logits[(size_t)i * vocab_size + word_index];
        sum += log(prob);
    }
    double avg_log_prob = sum / num_tokens;
    return float(exp(-avg_log_prob));
}

// get_dataset_perplexity perplexity.h:57-97 on token ids: tokens_with_bos[0] = BOS, targets follow.
float q4_perplexity_ids(Transformer* t, Sampler* sampler, const int* tokens_with_bos, int num_tokens) {
    Config* config = &t->config;
    RunState* state = &t->state;
    if (!state->logits_array || num_tokens < 1) return -1.0f;
    if (num_tokens >= config->seq_len) num_tokens = config->seq_len - 1;           // :68-72
    const unsigned long long rng0 = sampler->rng_state;
    for (int attempt = 0;; attempt++) {
        if (q4_reset_sequence(state, tokens_with_bos, num_tokens + 1)) return -1.0f;   // :76-78
        // the input tokens are all known: the steps are queued back to back (the device advances its own position) and the
        // host synchronises once, where the reference calls cudaDeviceSynchronize after every step (:81)
        for (int pos = 0; pos < num_tokens; pos++)
            if (q4_run_transformer_at(pos, 0, config, state, &t->weights, 1, sampler)) return -1.0f;   // :80
        if (hipDeviceSynchronize() != hipSuccess) return -1.0f;                        // :81
        if (!q4_handoff_status(state)) break;
        // a timed-out in-launch wait: the library has dropped to fusion level 1 and cleared its state; redo the pass once
        if (attempt > 0) return -1.0f;
        sampler->rng_state = rng0;
    }
    float* logits_arr = (float*)malloc((size_t)num_tokens * config->vocab_size * sizeof(float));
    if (q4_get_logits_array(t, num_tokens, logits_arr)) { free(logits_arr); return -1.0f; }   // :88-89
    float pplx = compute_perplexity(tokens_with_bos + 1, logits_arr, num_tokens, config->vocab_size);   // :91
    free(logits_arr);
    return pplx;
}

}  // extern "C"
