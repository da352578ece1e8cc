// q4_bench.hip -- per-kernel timing with dispatch timestamps over a ring of distinct weight sets (bench.py).
#include <hip/hip_runtime.h>
#include <time.h>
#include <vector>
#include "q4_internal.h"
using namespace q4;

// launch `kernel_id` on the i-th weight set of the ring (a different layer's weights every launch: > 256 MB in flight, past the
// Infinity Cache). public_attention: id 6 through the C-ABI entry point (caller-owned `att`, two-launch merge) instead of the
// network's internal launcher
static int launch_by_id(int kernel_id, int i, const Config* p, RunState* s, const TransformerWeights* w, bool public_attention = false) {
    const int dim = p->dim, hidden = p->hidden_dim;
    const int head_size = dim / p->n_heads;
    const int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    const PerLayerWeight* L = &w->layers[i % w->num_layers];   // ring: a different layer's weights every launch
    const long long loff = (long long)(i % w->num_layers) * p->seq_len * kv_dim;
    switch (kernel_id) {
        case 0: return launch_ffn_fused(s->hb, s->x, L->rms_ffn_weight, &L->wq_gate, &L->wq_up, dim, hidden);
        case 1: return q4_matmul_q4(s->hb, s->xb, &L->wq_gate, dim, hidden, 0, -1, nullptr);
        case 2: return q4_matmul_q4(s->xb, s->hb, &L->wq_down, hidden, dim, 1, -1, nullptr);
        case 3: return launch_qkv_fused(s->q, s->key_cache, s->value_cache, s->x, L->rms_att_weight, &L->wq_q, &L->wq_k,
                                        &L->wq_v, dim, kv_dim, loff, s->pos, head_size, p->rope_theta, rope_table_of(s), nullptr);
        case 4: return q4_matmul_q4(s->q, s->xb, &L->wq_o, dim, dim, 1, -1, nullptr);
        case 5: return q4_matmul_f16(s->logits, s->x, w->wcls, p->dim, p->vocab_size, 1, 0, 0, 0, -1, 1.0f);
        case 6:
            if (public_attention)
                return q4_multi_head_attention(s->xb, s->q, s->key_cache + loff, s->value_cache + loff, s->att, p->n_heads, head_size,
                                               p->n_heads / p->n_kv_heads, p->seq_len, s->pos);
            return launch_attention(s->xb, s->q, s->key_cache + loff, s->value_cache + loff, p->n_heads, head_size,
                                    p->n_heads / p->n_kv_heads, p->seq_len, s->pos, (float*)s->att, att_buffer_bytes(p), nullptr);
        case 7: return q4_rmsnorm(s->xb, s->x, w->rms_final_weight, dim);
        case 8: return q4_argmax(s->logits, p->vocab_size, &(s->shared_data->tokens[0]), &(s->shared_data->pos), s->pos, 0);
        case 9: return q4_copy_embedding(s->x, w->token_embedding_table, dim, s->shared_data->tokens, s->pos);
        case 10: {   // the FFN half of a layer as one launch (gemv_ffn_pair.h); no QKV launch advances the epoch here: distinct tags by launch index
            unsigned* sync = sync_words_of_state(s);
            if (!sync) return Q4_ERR_ARG;
            return launch_ffn_pair(s->x, s->hb, L->rms_ffn_weight, &L->wq_gate, &L->wq_up, &L->wq_down, dim, hidden, sync, ffn_pair_sync_offset(dim), 1u + (unsigned)i);
        }
        case 11: {   // ... with the next layer's QKV as its third phase (fusion level 5)
            unsigned* sync = sync_words_of_state(s);
            if (!sync) return Q4_ERR_ARG;
            const PerLayerWeight* N = &w->layers[(i + 1) % w->num_layers];
            const long long noff = (long long)((i + 1) % w->num_layers) * p->seq_len * kv_dim;
            const FfnQkvNext nx = {N->rms_att_weight, &N->wq_q, &N->wq_k, &N->wq_v, s->q, s->key_cache + noff, s->value_cache + noff, s->pos, rope_table_of(s), nullptr, kv_dim, head_size};
            return launch_ffn_pair(s->x, s->hb, L->rms_ffn_weight, &L->wq_gate, &L->wq_up, &L->wq_down, dim, hidden, sync, ffn_pair_sync_offset(dim), 1u + (unsigned)i, &nx);
        }
    }
    return Q4_ERR_ARG;
}

// kernel 10 leaves granules tagged beyond the model's epoch behind: cleared, so that no later launch of the network can meet one of them as its own
static void forget_bench_granules(int kernel_id, const Config* p, RunState* s) {
    unsigned* sync = sync_words_of_state(s);
    if ((kernel_id != 10 && kernel_id != 11) || !sync) return;
    (void)hipMemsetAsync(sync + ffn_pair_sync_offset(p->dim), 0, ffn_pair_sync_words(p->dim, p->hidden_dim) * sizeof(unsigned), g_stream);
    (void)hipStreamSynchronize(g_stream);
}

// Steady-state cost of one launch INSIDE a hipGraph (what the decode loop pays: kernel + boundary): `iters` launches
// of kernel_id over the ring of layers are captured into one graph, replayed `reps` times; returns microseconds per
// launch from the wall clock around the replays (after one warm replay). <0 on error.
extern "C" double q4_bench_kernel_graph(int kernel_id, const Config* p, RunState* s, const TransformerWeights* w, int iters,
                                        int reps) {
    if (iters < 1 || reps < 1 || !g_stream) return -1.0;
    if (s->shared_data->pos >= p->seq_len) return -1.0;   // the kernels write the KV row of the device position: it must exist
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(g_stream, hipStreamCaptureModeGlobal) != hipSuccess) return -1.0;
    int rc = 0;
    for (int i = 0; i < iters && !rc; i++) rc = launch_by_id(kernel_id, i, p, s, w);
    if (hipStreamEndCapture(g_stream, &graph) != hipSuccess || rc) { if (graph) hipGraphDestroy(graph); return -1.0; }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { hipGraphDestroy(graph); return -1.0; }
    hipGraphDestroy(graph);
    double us = -1.0;
    if (hipGraphLaunch(exec, g_stream) == hipSuccess && hipStreamSynchronize(g_stream) == hipSuccess) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        bool ok = true;
        for (int r = 0; r < reps && ok; r++) ok = hipGraphLaunch(exec, g_stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(g_stream) == hipSuccess;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (ok) us = ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3) / ((double)iters * reps);
    }
    hipGraphExecDestroy(exec);
    forget_bench_granules(kernel_id, p, s);
    return us;
}

extern "C" double q4_bench_kernel(int kernel_id, const Config* p, RunState* s, const TransformerWeights* w, int iters,
                                  double* min_us, double* max_us) {
    if (iters < 1 || !p || !s || !w) return -1.0;
    if (s->shared_data->pos >= p->seq_len) return -1.0;   // the kernels write the KV row of the device position: it must exist
    if ((kernel_id < 0 || kernel_id > 6) && kernel_id != 10 && kernel_id != 11) return -1.0;       // ids 7-9 (tiny launches) are timed inside a graph only
    std::vector<hipEvent_t> ev(2 * iters);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return -1.0;
    int rc = 0;
    for (int i = 0; i < iters && !rc; i++) {
        g_ev_start = ev[2 * i];
        g_ev_stop = ev[2 * i + 1];
        rc = launch_by_id(kernel_id, i, p, s, w, true);
    }
    g_ev_start = g_ev_stop = nullptr;
    double total = 0, mn = 1e30, mx = 0;
    if (!rc && hipStreamSynchronize(g_stream) != hipSuccess) rc = Q4_ERR_HIP;
    for (int i = 0; i < iters && !rc; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { rc = Q4_ERR_HIP; break; }
        const double us = ms * 1000.0;
        total += us;
        if (us < mn) mn = us;
        if (us > mx) mx = us;
    }
    for (auto& e : ev) hipEventDestroy(e);
    forget_bench_granules(kernel_id, p, s);
    if (rc) return -1.0;
    if (min_us) *min_us = mn;
    if (max_us) *max_us = mx;
    return total / iters;
}
