// q4_host.cpp -- host-only side of llama2_q4: the llama2.c BPE tokenizer (tokenizer.h), the generate / chat /
// perplexity drivers (llama2_q4.cu:436-601, perplexity.h:57-139) and the CLI (llama2_q4.cu:604-720).
// No device code here; everything reaches the GPU through the C ABI in llama2_q4.h.
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>
#include "llama2_q4.h"

static const int bos_token = 1;   // tokenizer.h:8
static const int eos_token = 2;   // tokenizer.h:9

// ---------------------------------------------------------------------------------------------------
// Tokenizer (tokenizer.h:16-28). Lookup uses the same libc qsort/bsearch pair as the reference so that
// duplicate vocabulary strings (raw-byte pieces at ids 3..258) resolve to the same id.
struct TokenIndex {
    const char* str;
    int id;
};
struct Tokenizer {
    std::vector<std::string> vocab;
    std::vector<float> vocab_scores;
    std::vector<TokenIndex> sorted_vocab;   // built lazily by encode()
    int vocab_size;
    unsigned int max_token_length;
    unsigned char byte_pieces[512];
};

static int compare_tokens(const void* a, const void* b) {
    return strcmp(((const TokenIndex*)a)->str, ((const TokenIndex*)b)->str);
}

static int str_lookup(const char* str, Tokenizer* t) {                             // tokenizer.h:95-100
    TokenIndex tok = {str, 0};
    TokenIndex* res = (TokenIndex*)bsearch(&tok, t->sorted_vocab.data(), t->vocab_size, sizeof(TokenIndex), compare_tokens);
    return res != NULL ? res->id : -1;
}

extern "C" {

struct Tokenizer* q4_tokenizer_new(const char* tokenizer_path, int vocab_size) {   // build_tokenizer :35-59
    FILE* file = fopen(tokenizer_path, "rb");
    if (!file) { fprintf(stderr, "couldn't load %s\n", tokenizer_path); return nullptr; }
    Tokenizer* t = new Tokenizer();
    t->vocab_size = vocab_size;
    t->vocab.resize(vocab_size);
    t->vocab_scores.resize(vocab_size);
    for (int i = 0; i < 256; i++) {
        t->byte_pieces[i * 2] = (unsigned char)i;
        t->byte_pieces[i * 2 + 1] = '\0';
    }
    bool ok = fread(&t->max_token_length, sizeof(int), 1, file) == 1;
    for (int i = 0; ok && i < vocab_size; i++) {
        int len = 0;
        ok = fread(&t->vocab_scores[i], sizeof(float), 1, file) == 1 && fread(&len, sizeof(int), 1, file) == 1 && len >= 0;
        if (!ok) break;
        t->vocab[i].resize(len);
        ok = len == 0 || fread(&t->vocab[i][0], len, 1, file) == 1;
    }
    fclose(file);
    if (!ok) { fprintf(stderr, "failed read\n"); delete t; return nullptr; }
    return t;
}

void q4_tokenizer_delete(struct Tokenizer* t) { delete t; }
int q4_tokenizer_max_token_length(const struct Tokenizer* t) { return (int)t->max_token_length; }

const char* q4_tokenizer_decode(struct Tokenizer* t, int prev_token, int token) {  // decode :68-79
    const char* piece = t->vocab[token].c_str();
    if (prev_token == bos_token && piece[0] == ' ') piece++;
    unsigned char byte_val;
    if (sscanf(piece, "<0x%02hhX>", &byte_val) == 1) piece = (const char*)t->byte_pieces + byte_val * 2;
    return piece;
}

static void safe_printf(const char* piece) {                                       // tokenizer.h:81-93
    if (piece == NULL || piece[0] == '\0') return;
    if (piece[1] == '\0') {
        unsigned char byte_val = piece[0];
        if (!(isprint(byte_val) || isspace(byte_val))) return;
    }
    printf("%s", piece);
}

int q4_tokenizer_encode(struct Tokenizer* t, const char* text, int bos, int eos, int* tokens, int* n_tokens) {   // encode :102-223
    if (text == NULL) { fprintf(stderr, "cannot encode NULL text\n"); return Q4_ERR_ARG; }
    if (t->sorted_vocab.empty()) {
        t->sorted_vocab.resize(t->vocab_size);
        for (int i = 0; i < t->vocab_size; i++) {
            t->sorted_vocab[i].str = t->vocab[i].c_str();
            t->sorted_vocab[i].id = i;
        }
        qsort(t->sorted_vocab.data(), t->vocab_size, sizeof(TokenIndex), compare_tokens);
    }
    std::vector<char> buf(t->max_token_length * 2 + 1 + 2 + 8);
    char* str_buffer = buf.data();
    size_t str_len = 0;
    *n_tokens = 0;
    if (bos) tokens[(*n_tokens)++] = bos_token;
    if (text[0] != '\0') {                                                         // dummy prefix, :132-136
        int dummy_prefix = str_lookup(" ", t);
        tokens[(*n_tokens)++] = dummy_prefix;
    }
    for (const char* c = text; *c != '\0'; c++) {
        if ((*c & 0xC0) != 0x80) str_len = 0;                                      // not a continuation byte
        str_buffer[str_len++] = *c;
        str_buffer[str_len] = '\0';
        if ((*(c + 1) & 0xC0) == 0x80 && str_len < 4) continue;
        int id = str_lookup(str_buffer, t);
        if (id != -1) {
            tokens[(*n_tokens)++] = id;
        } else {
            for (size_t i = 0; i < str_len; i++) tokens[(*n_tokens)++] = (unsigned char)str_buffer[i] + 3;   // byte fallback
        }
        str_len = 0;
    }
    std::string merged;
    while (1) {                                                                    // greedy best-score merges :186-217
        float best_score = -1e10;
        int best_id = -1, best_idx = -1;
        for (int i = 0; i < (*n_tokens - 1); i++) {
            merged = t->vocab[tokens[i]];
            merged += t->vocab[tokens[i + 1]];
            int id = str_lookup(merged.c_str(), t);
            if (id != -1 && t->vocab_scores[id] > best_score) {
                best_score = t->vocab_scores[id];
                best_id = id;
                best_idx = i;
            }
        }
        if (best_idx == -1) break;
        tokens[best_idx] = best_id;
        for (int i = best_idx + 1; i < (*n_tokens - 1); i++) tokens[i] = tokens[i + 1];
        (*n_tokens)--;
    }
    if (eos) tokens[(*n_tokens)++] = eos_token;
    return Q4_OK;
}

// ---------------------------------------------------------------------------------------------------
static long time_in_ms() {                                                         // llama2_q4.cu:400-405
    struct timespec time;
    timespec_get(&time, TIME_UTC);
    return time.tv_sec * 1000 + time.tv_nsec / 1000000;
}

static void die_on(int rc) {
    if (rc) { printf("\n%s\n", q4_status_string(rc)); exit(EXIT_FAILURE); }        // the reference's printf + exit
}

// generate(), llama2_q4.cu:436-492
double q4_generate(Transformer* transformer, struct Tokenizer* tokenizer, Sampler* sampler, const char* prompt, int steps,
                   int* timed_tokens_out, double* seconds_out) {
    if (prompt == NULL) prompt = "";
    int num_prompt_tokens = 0;
    int* prompt_tokens = (int*)malloc((strlen(prompt) + 3) * sizeof(int));
    printf("\nEncoding Prompt... ");
    q4_tokenizer_encode(tokenizer, prompt, 1, 0, prompt_tokens, &num_prompt_tokens);
    printf("Done!\n");
    if (num_prompt_tokens < 1) {
        fprintf(stderr, "something is wrong, expected at least 1 prompt token\n");
        exit(EXIT_FAILURE);
    }
    RunState* state = &transformer->state;
    const unsigned long long rng0 = sampler->rng_state;
    int timed_tokens = 0;
    double time = 0.0;
    for (int attempt = 0;; attempt++) {
        long start = time_in_ms();
        int next;
        int token = prompt_tokens[0];
        int pos = 0;
        die_on(q4_reset_sequence(state, prompt_tokens, num_prompt_tokens));            // :461-463
        int queued = 0, group_start = 0;
        unsigned long long group_rng = sampler->rng_state;
        bool stopped = false;
        while (pos < steps) {
            // step `pos` is queued behind step pos-1 before the host waits for step pos-1's token (reference: sync, then
            // launch, :468-470) -- same device order, the GPU never idles between tokens; greedy steps inside one bin go out
            // Q4_MULTI_STEPS at a time
            if (pos >= queued) {
                const int k = q4_steps_that_fit(pos, num_prompt_tokens, steps, &transformer->config, sampler);
                group_start = pos;
                group_rng = sampler->rng_state;
                die_on(q4_run_transformer_steps(pos, k, pos >= num_prompt_tokens - 1, &transformer->config, state, &transformer->weights, 0, sampler));
                queued = pos + k;
            }
            die_on(q4_wait_pos(state, pos));                                           // :468
            if (pos > 0) {
                next = q4_shared_token(state, pos);                                    // output token of the previous iteration
                if (next >= transformer->config.vocab_size) next = 0;                  // :474
                const char* piece = q4_tokenizer_decode(tokenizer, token, next);
                safe_printf(piece);
                if (next == eos_token) { stopped = true; break; }
                token = next;
            }
            pos++;
        }
        printf("\n");
        long end = time_in_ms();
        if (stopped) {   // one coin per executed step as in the reference (sampler.h:45): undo the draws of the group's surplus steps
            sampler->rng_state = group_rng;
            for (int i = group_start; i <= pos; i++) (void)random_f32(&sampler->rng_state);
        }
        time = (end - start) / 1000.0;
        timed_tokens = pos - 1;
        if (q4_handoff_status(state) == Q4_OK) break;
        // a bounded in-launch wait ran out: the text above is invalid. The library has cleared its hand-off state and dropped
        // to fusion level 1 (no in-launch waits): say so and generate again, once
        if (attempt > 0) die_on(Q4_ERR_HIP);
        printf("\n[%s -- generating again]\n", q4_last_error());
        sampler->rng_state = rng0;
    }
    printf("\nachieved tok/s: %f. Tokens: %d, seconds: %g\n", timed_tokens / time, timed_tokens, time);   // :489
    free(prompt_tokens);
    if (timed_tokens_out) *timed_tokens_out = timed_tokens;
    if (seconds_out) *seconds_out = time;
    return timed_tokens / time;
}

static void read_stdin(const char* guide, char* buffer, size_t bufsize) {          // :494-503
    printf("%s", guide);
    if (fgets(buffer, bufsize, stdin) != NULL) {
        size_t len = strlen(buffer);
        if (len > 0 && buffer[len - 1] == '\n') buffer[len - 1] = '\0';
    } else {
        buffer[0] = '\0';
    }
}

// chat(), llama2_q4.cu:507-601
void q4_chat(Transformer* transformer, struct Tokenizer* tokenizer, Sampler* sampler, const char* cli_user_prompt,
             const char* cli_system_prompt, int steps) {
    char system_prompt[512];
    char user_prompt[512];
    char rendered_prompt[1152];
    int num_prompt_tokens = 0;
    int* prompt_tokens = (int*)malloc(1152 * sizeof(int));
    int user_idx = 0;
    int8_t user_turn = 1;
    int next = 0;
    int token = 0;
    int pos = 0;
    RunState* state = &transformer->state;
    system_prompt[0] = '\0';
    die_on(q4_reset_sequence(state, nullptr, 0));                                  // :526-527
    while (pos < steps) {
        if (user_turn) {
            if (feof(stdin)) break;   // scripted stdin ran dry (the reference would spin on empty prompts)
            if (pos == 0) {
                if (cli_system_prompt == NULL) read_stdin("Enter system prompt (optional): ", system_prompt, sizeof(system_prompt));
                else { strncpy(system_prompt, cli_system_prompt, sizeof(system_prompt) - 1); system_prompt[sizeof(system_prompt) - 1] = 0; }
            }
            if (pos == 0 && cli_user_prompt != NULL) { strncpy(user_prompt, cli_user_prompt, sizeof(user_prompt) - 1); user_prompt[sizeof(user_prompt) - 1] = 0; }
            else read_stdin("User: ", user_prompt, sizeof(user_prompt));
            if (pos == 0 && system_prompt[0] != '\0')
                snprintf(rendered_prompt, sizeof(rendered_prompt), "[INST] <<SYS>>\n%s\n<</SYS>>\n\n%s [/INST]", system_prompt, user_prompt);
            else
                snprintf(rendered_prompt, sizeof(rendered_prompt), "[INST] %s [/INST]", user_prompt);
            printf("\nRendered prompt: %s\n", rendered_prompt);                    // :564
            q4_tokenizer_encode(tokenizer, rendered_prompt, 1, 0, prompt_tokens, &num_prompt_tokens);
            user_idx = 0;
            user_turn = 0;
            printf("Assistant: ");
            die_on(q4_stream_synchronize());
            for (int i = 0; i < num_prompt_tokens && pos + i < Q4_MAX_SEQ_LEN; i++)
                transformer->state.shared_data->tokens[pos + i] = prompt_tokens[i];   // :573
        }
        die_on(q4_stream_synchronize());                                           // :578
        die_on(q4_run_transformer(user_idx >= num_prompt_tokens - 1, &transformer->config, state, &transformer->weights, 0, sampler));
        user_idx++;
        if (user_idx > 0) {
            next = q4_shared_token(state, pos);                                    // :584
            if (next == eos_token) {
                user_turn = 1;
                printf("\n");
                if (q4_handoff_status(state)) {   // the turn just printed came out of a timed-out in-launch wait: say so, stop
                    printf("\n%s\n", q4_last_error());
                    break;
                }
            } else if (user_idx > num_prompt_tokens) {
                const char* piece = q4_tokenizer_decode(tokenizer, token, next);
                safe_printf(piece);
            }
            token = next;
        }
        pos++;
    }
    printf("\n");
    if (q4_handoff_status(state)) printf("\n%s\n", q4_last_error());
    free(prompt_tokens);
}

// get_dataset_perplexity, perplexity.h:57-97
float q4_get_dataset_perplexity(char* dataset, struct Tokenizer* tokenizer, Transformer* t, Sampler* sampler) {
    int bytes = strlen(dataset);
    Config* config = &t->config;
    std::vector<int> toks(bytes + 4);
    printf("\nTokenizing Dataset...");
    int totalTokens = 0;
    toks[0] = bos_token;                                                           // :78
    q4_tokenizer_encode(tokenizer, dataset, 0, 0, toks.data() + 1, &totalTokens);  // :63
    printf("done!\n");
    printf("Found %d characters, %d tokens", bytes, totalTokens);
    int numTokens = totalTokens;
    if (numTokens >= config->seq_len) {
        numTokens = config->seq_len - 1;
        printf("\nTruncated to %d tokens", numTokens);
    }
    printf("\nRunning the network to get logits...");
    float pplx = q4_perplexity_ids(t, sampler, toks.data(), numTokens);            // :74-91
    printf("done!\n");
    printf("Computing perplexity...");
    printf("\nPerplexity computed on %d tokens: %f\n\n", numTokens, pplx);         // :93
    return pplx;
}

// parseDataSetAndComputePreplexity, perplexity.h:99-139
double q4_parse_dataset_and_compute_perplexity(const char* textFileName, struct Tokenizer* tokenizer, Transformer* t,
                                               Sampler* sampler) {
    FILE* fp = fopen(textFileName, "rb");
    if (!fp) { printf("Couldn't open file %s\n", textFileName ? textFileName : "(null)"); return -1.0; }
    printf("\nLoading Dataset...");
    fseek(fp, 0, SEEK_END);
    long bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char* dataset = (char*)malloc(bytes + 1);
    if (fread(dataset, 1, bytes, fp) != (size_t)bytes) bytes = 0;
    fclose(fp);
    printf("done!\n");
    dataset[bytes] = 0;
    int count = 0;
    double pplx_product = 1;
    char* currentSeq = dataset;
    while (currentSeq) {
        char* nextseq = strstr(currentSeq, "<|endoftext|>");
        if (nextseq) {
            *nextseq = 0;
            nextseq += 13;
            pplx_product *= q4_get_dataset_perplexity(currentSeq, tokenizer, t, sampler);
            count++;
            currentSeq = nextseq;
        } else {
            pplx_product *= q4_get_dataset_perplexity(currentSeq, tokenizer, t, sampler);
            count++;
            break;
        }
    }
    free(dataset);
    double geo = pow(pplx_product, 1.0 / count);
    printf("\nGeomean perplexity on %d sequences: %f\n\n", count, geo);            // :138
    return geo;
}

// ---------------------------------------------------------------------------------------------------
// CLI, llama2_q4.cu:604-720
static void error_usage_text(const char* argv0) {
    fprintf(stderr, "Usage:   %s <checkpoint> [options]\n", argv0);
    fprintf(stderr, "Example: %s model.bin -n 256 -i \"Write a poem on GPUs\"\n", argv0);
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "  -n <int>    max number of steps to run for, default = max_seq_len\n");
    fprintf(stderr, "  -i <string> input prompt\n");
    fprintf(stderr, "  -f <string> path to file containing input prompt. Can be used with for multi-line prompts.\n");
    fprintf(stderr, "  -t <float>  temperature in [0,inf], default 0.5\n");
    fprintf(stderr, "  -p <float>  p value in top-p (nucleus) sampling in [0,1] default 0.9\n");
    fprintf(stderr, "  -s <int>    random seed, default time(NULL)\n");
    fprintf(stderr, "  -z <string> optional path to custom tokenizer\n");
    fprintf(stderr, "  -m <string> mode: generate|chat|perplexity, default: generate\n");
    fprintf(stderr, "  -y <string> (optional) system prompt in chat mode\n");
    fprintf(stderr, "  -q <string> dataset file for computing perplexity\n");
}

static std::string g_file_prompt;

int q4_parse_args(int argc, char** argv, q4_cli_args* a) {
    // defaults, llama2_q4.cu:624-637
    a->checkpoint_path = NULL;
    a->tokenizer_path = "tokenizer.bin";
    a->dataset_path = NULL;
    a->steps = 0;
    a->prompt = NULL;
    a->perplexity = 0;
    a->temperature = 0.5f;
    a->topp = 0.6f;          // the code default (help text says 0.9)
    a->rng_seed = 0;
    a->mode = "generate";
    a->system_prompt = NULL;
    a->seed_from_time = 0;
    if (argc >= 2) a->checkpoint_path = argv[1]; else return 1;
    for (int i = 2; i < argc; i += 2) {
        if (i + 1 >= argc) return 1;
        if (argv[i][0] != '-') return 1;
        if (strlen(argv[i]) != 2) return 1;
        switch (argv[i][1]) {
            case 'n': a->steps = atoi(argv[i + 1]); break;
            case 'i': a->prompt = argv[i + 1]; break;
            case 'z': a->tokenizer_path = argv[i + 1]; break;
            case 't': a->temperature = atof(argv[i + 1]); break;
            case 'p': a->topp = atof(argv[i + 1]); break;
            case 's': a->rng_seed = atoi(argv[i + 1]); break;                     // atoi into 64 bits, P9
            case 'm': a->mode = argv[i + 1]; break;
            case 'y': a->system_prompt = argv[i + 1]; break;
            case 'q': a->dataset_path = argv[i + 1]; break;
            case 'f': {
                FILE* file = fopen(argv[i + 1], "r");
                if (!file) { printf("Couldn't open file %s\n", argv[i + 1]); exit(1); }
                fseek(file, 0, SEEK_END);
                long fsize = ftell(file);
                fseek(file, 0, SEEK_SET);
                if (a->prompt) printf("Warning: -f overrides -i\n");
                g_file_prompt.resize(fsize);
                if (fsize > 0 && fread(&g_file_prompt[0], fsize, 1, file) != 1) g_file_prompt.clear();
                fclose(file);
                a->prompt = g_file_prompt.c_str();
                break;
            }
            default: return 1;
        }
    }
    if (strcmp(a->mode, "perplexity") == 0) a->perplexity = 1;                     // :678
    // parameter validation/overrides :681-685
    if (a->rng_seed <= 0) { a->rng_seed = (unsigned int)time(NULL); a->seed_from_time = 1; }
    if (a->temperature < 0.0) a->temperature = 0.0;
    if (a->topp < 0.0 || 1.0 < a->topp) a->topp = 0.9;
    return 0;
}

int q4_main(int argc, char** argv) {
    q4_cli_args a;
    if (q4_parse_args(argc, argv, &a)) { error_usage_text(argv[0]); exit(EXIT_FAILURE); }
    if (!a.perplexity && a.dataset_path) printf("Warning: dataset path is ignored in non-perplexity mode\n");

    Transformer transformer;
    int rc = q4_build_transformer(&transformer, a.checkpoint_path, a.perplexity);
    if (rc) exit(rc == Q4_ERR_IO ? 1 : EXIT_FAILURE);
    int steps = a.steps;
    if (steps <= 0 || steps > transformer.config.seq_len) steps = transformer.config.seq_len;   // :690

    struct Tokenizer* tokenizer = q4_tokenizer_new(a.tokenizer_path, transformer.config.vocab_size);
    if (!tokenizer) exit(EXIT_FAILURE);
    Sampler sampler;
    die_on(build_sampler(&sampler, transformer.config.vocab_size, a.temperature, a.topp, a.rng_seed));

    q4_stream_t stream;
    die_on(q4_stream_create(&stream));                                             // :700
    q4_set_stream(stream);

    if (a.perplexity) {
        q4_parse_dataset_and_compute_perplexity(a.dataset_path, tokenizer, &transformer, &sampler);
    } else if (strcmp(a.mode, "generate") == 0) {
        q4_generate(&transformer, tokenizer, &sampler, a.prompt, steps, nullptr, nullptr);
    } else if (strcmp(a.mode, "chat") == 0) {
        q4_chat(&transformer, tokenizer, &sampler, a.prompt, a.system_prompt, steps);
    } else {
        error_usage_text(argv[0]);
        exit(EXIT_FAILURE);
    }
    q4_stream_synchronize();
    q4_reset_graphs();                                                             // :713-716
    q4_free_transformer(&transformer);
    destroy_sampler(&sampler);
    q4_set_stream(nullptr);
    q4_stream_destroy(stream);
    q4_tokenizer_delete(tokenizer);
    return 0;
}

}  // extern "C"
