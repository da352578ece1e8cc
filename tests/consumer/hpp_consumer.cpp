// A compiled consumer of include/llama2_q4.hpp: the reference's host-function names (llama2_q4.cu:209-432) over the C ABI,
// used the way a fork of llama2_q4.cu would use them. Built and run by tests/test_abi.py (no arguments: link check only)
// and tests/test_consumer_gpu.py (with checkpoints: decode through run_transformer, free, rebuild a DIFFERENT model on
// the same Transformer object without a manual q4_reset_graphs -- the stale-graph case of free_transformer).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "llama2_q4.hpp"

using namespace llama2_q4;

static void decode(Transformer* t, Sampler* smp, const int* prompt, int nprompt, int steps, const char* tag) {
    Config* p = &t->config;
    RunState* s = &t->state;
    die_on(q4_reset_sequence(s, prompt, nprompt));
    for (int pos = 0; pos < steps; pos++) {
        die_on(q4_stream_synchronize());                                         // llama2_q4.cu:468
        run_transformer(pos >= nprompt - 1, p, s, &t->weights, false, smp);      // :470
    }
    die_on(q4_stream_synchronize());
    printf("%s tokens", tag);
    for (int i = 0; i <= steps; i++) printf(" %d", s->shared_data->tokens[i]);
    std::vector<q4_half> logits(p->vocab_size);
    die_on(q4_get_logits(t, logits.data()));
    unsigned long long sum = 0;
    for (int i = 0; i < p->vocab_size; i++) sum = sum * 1000003ull + logits[i];
    printf("\n%s logits_hash %llu pos %d\n", tag, sum, s->shared_data->pos);
}

int main(int argc, char** argv) {
    if (argc < 2) {
        for (int rc = 0; rc <= 5; rc++) printf("%d %s\n", rc, q4_status_string(rc));
        // take the address of every wrapper so that the link step resolves all of them
        void* fns[] = {(void*)&rmsnorm, (void*)(void (*)(half_t*, half_t*, half_t*, int, int, int, int, int, int, int, float))&matmul,
                       (void*)(void (*)(half_t*, half_t*, QWeight&, int, int, bool, int, int*))&matmul, (void*)&qkv_matvec,
                       (void*)&ffn_matvec_silu, (void*)&RoPERotation, (void*)&MultiHeadAttention, (void*)&run_llama_network,
                       (void*)&run_transformer, (void*)&sample, (void*)&build_transformer, (void*)&free_transformer};
        printf("wrappers %zu\n", sizeof(fns) / sizeof(fns[0]));
        return 0;
    }
    q4_set_quiet(1);
    die_on(q4_set_device(0));
    q4_stream_t stream;
    die_on(q4_stream_create(&stream));                                           // cudaStreamCreate, :700
    q4_set_stream(stream);
    const int prompt[4] = {1, 20, 300, 45};
    Transformer t;                                                               // a stack object: same address for both models
    Sampler smp;
    for (int m = 1; m < argc; m++) {
        build_transformer(&t, argv[m], false);                                   // :408
        die_on(build_sampler(&smp, t.config.vocab_size, 0.0f, 0.9f, 1));
        char tag[32];
        snprintf(tag, sizeof(tag), "model%d", m);
        decode(&t, &smp, prompt, 4, 10, tag);
        if (m == 1) {
            // the per-kernel layer under the reference's names: one unfused attention-block prefix on layer 0
            Config* p = &t.config;
            RunState* s = &t.state;
            PerLayerWeight* L = &t.weights.layers[0];
            const int kv_dim = p->dim * p->n_kv_heads / p->n_heads;
            rmsnorm(s->xb, s->x, L->rms_att_weight, p->dim);                     // :300
            if (kv_dim == p->dim) qkv_matvec(s->q, s->key_cache, s->value_cache, s->xb, L->wq_q, L->wq_k, L->wq_v, p->dim, p->dim, 0, s->pos);
            matmul(s->xb, s->q, L->wq_o, p->dim, p->dim);                        // plain int4 GEMV
            matmul(s->logits, s->x, t.weights.wcls, p->dim, p->vocab_size);      // fp16 GEMV
            die_on(q4_stream_synchronize());
            printf("kernels ok\n");
        }
        destroy_sampler(&smp);
        free_transformer(&t);                                                    // :428 -- must drop this model's captured graphs
    }
    q4_set_stream(nullptr);
    die_on(q4_stream_destroy(stream));
    return 0;
}
