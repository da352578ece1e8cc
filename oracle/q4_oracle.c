/*
 * q4_oracle.c -- CPU restatement of the llama2_q4 decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under llama_cu_awq_amd/ (the product) may
 * include, link, load or execute this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED for the device kernels: the reference (ankan-ban/llama_cu_awq)
 * ships no tests, no golden vectors and no CPU forward(); its CUDA path cannot be
 * compiled or run in the build container (no nvcc, no NVIDIA GPU, inline PTX + cub).
 * This file therefore restates the reference's arithmetic line by line from its
 * sources; every function cites the reference file:line it follows.  The pieces of
 * the reference that DO compile here (weight_packer.cpp, tokenizer.h) are built
 * from /root/reference by oracle/Makefile into oracle/_ref/ and pin the .bin
 * format and the tokenizer goldens (tests/golden/).
 *
 * Rounding model: every buffer between kernels is fp16,
 * all arithmetic inside a kernel is fp32 with fused multiply-add where nvcc's
 * default -fmad=true contracts `a += b*c` (restated with explicit fmaf; compile this
 * file with -ffp-contract=off so nothing else is contracted).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint16_t f16;

/* ---- reference common.h:9-18: the 32-byte file header ---------------------- */
typedef struct {
    int dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len;
    float rope_theta;
} OrcConfig;

/* reference common.h:20-24 (host pointers here) */
typedef struct {
    const uint32_t *weight;
    const uint32_t *zeros;
    const f16 *scales;
} OrcQWeight;

typedef struct {
    const f16 *rms_att, *rms_ffn;
    OrcQWeight q, k, v, o, gate, up, down;
} OrcLayer;

typedef struct {
    OrcConfig cfg;
    uint8_t *blob;       /* whole .bin file */
    size_t blob_bytes;
    const f16 *embed, *wcls, *rms_final;
    OrcLayer *layers;
    /* run state, reference common.h:56-72 */
    f16 *x, *xb, *hb, *q, *att, *logits, *key_cache, *value_cache;
    int pos;
    int *tokens;
    /* unrounded evaluation (orc_forward_f64): double KV cache for the first f64_cap positions */
    double *k64, *v64;
    int f64_cap;
    /* tools/error_growth.py: the residual stream after the attention half (o-proj + residual) and after the FFN half (down + residual)
     * of every layer of the LAST forward, [n_layers][2][dim]; null = off */
    f16 *dump16;
    double *dump64;
} OrcModel;

/* ---- fp16 <-> fp32, exact IEEE binary16, round-to-nearest-even ------------- */
static float g_h2f[65536];
static int g_h2f_ready = 0;

static float h2f_slow(f16 h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        u = s | 0x7f800000u | (m << 13);
    } else {
        u = s | ((e + 112) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void orc_init(void) {
    if (g_h2f_ready) return;
    for (uint32_t i = 0; i < 65536; i++) g_h2f[i] = h2f_slow((f16)i);
    g_h2f_ready = 1;
}

static inline float h2f(f16 h) { return g_h2f[h]; }

static inline f16 f2h(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (f16)(s | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0)); /* inf / nan */
    if (a >= 0x477ff000u) return (f16)(s | 0x7c00u);      /* rounds to inf (>= 65520) */
    if (a < 0x33000001u) return (f16)s;                   /* < 2^-25 (or == 2^-25): rounds to 0 */
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;             /* 24-bit significand */
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;        /* bits to drop */
    uint32_t half_m = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1))) half_m++;
    uint32_t r;
    if (e < -14) r = half_m;                              /* subnormal (may carry into normal) */
    else r = ((uint32_t)(e + 15) << 10) + (half_m - 0x400u);  /* carry propagates into exponent */
    return (f16)(s | r);
}

float orc_h2f(f16 h) { orc_init(); return h2f(h); }
f16 orc_f2h(float f) { return f2h(f); }

static inline int divUp(int a, int b) { return (a - 1) / b + 1; }   /* reference common.h:80-82 */

/* reference llama2_q4.cu:82-98 */
size_t orc_qweight_sizes(int height, int width, size_t *w_u32, size_t *z_u32, size_t *s_f16) {
    size_t pwh = (size_t)divUp(height, 32) * 4;
    size_t sh = (size_t)divUp(height, 128);
    size_t pzh = (size_t)divUp((int)sh, 8);
    if (w_u32) *w_u32 = pwh * width;
    if (z_u32) *z_u32 = pzh * width;
    if (s_f16) *s_f16 = sh * width;
    return pwh * width * 4 + pzh * width * 4 + sh * width * 2;
}

/* ---- cub-like reductions (order only matters at fp32 rounding level) -------- */
/* shfl_down tree, offsets 1,2,4,8,16 (cub WarpReduceShfl); result = lane 0 */
static float warp_sum32(float *v) {
    for (int off = 1; off < 32; off <<= 1)
        for (int l = 0; l + off < 32; l += 2 * off) v[l] = v[l] + v[l + off];
    return v[0];
}

/* cub::BlockReduce<float,1024> (warp reductions, then thread 0 adds the 32 warp
   aggregates in order) */
static float block_sum1024(float *v) {
    float agg[32];
    for (int w = 0; w < 32; w++) agg[w] = warp_sum32(v + 32 * w);
    float s = agg[0];
    for (int w = 1; w < 32; w++) s = s + agg[w];
    return s;
}

/* ---- a2: get_mat_vec_int4, reference gpu_kernels.h:171-210 ------------------ */
/* lane-partitioned fp32 order exactly as the reference's 32-lane warp */
/* tools/error_growth.py control: 1 = another VALID fp32 order of the same sums (each lane walks its groups from the last to the first, the 32
 * lane sums are added sequentially instead of by the shuffle tree) -- how far apart do two legitimate orders of the reference's own arithmetic
 * end up after 32 layers? 0 (always, outside that tool) = the reference's order. */
static int g_order_variant = 0;
void orc_set_order_variant(int v) { g_order_variant = v; }

float orc_dot_q4(int index, const f16 *input, const uint32_t *qw, const uint32_t *qz, const f16 *sc,
                 int inputElements, int pzh, int sh, int pwh) {
    float lane[32];
    for (int tx = 0; tx < 32; tx++) {
        float sum = 0.f;
        int nq = 0;
        while (nq * 128 + tx * 4 < pwh) nq++;
        for (int it = 0; it < nq; it++) {                                     /* :176 */
            const int ygq = g_order_variant ? nq - 1 - it : it;
            uint32_t packed_q_z = qz[(size_t)index * pzh + ygq];              /* :177 */
            const uint32_t *lw = &qw[(size_t)index * pwh + ygq * 128 + tx * 4]; /* :181 */
            int group_y = ygq * 8 + (tx / 4);                                 /* :183 */
            float q_z = (float)((packed_q_z >> (4 * (tx / 4))) & 0xF);        /* :184 */
            float scale = h2f(sc[(size_t)index * sh + group_y]);              /* :185 */
            int y_base = ygq * 1024 + tx * 32;                                /* :186 */
            for (int qi = 0; qi < 4; qi++) {                                  /* :188 */
                int ys = y_base + qi * 8;
                if (ys < inputElements) {                                     /* :190 */
                    uint32_t packed_q_w = lw[qi];
                    for (int i = 0; i < 8; i++) {                             /* :195-200 */
                        float q_wt = (float)(packed_q_w & 0xF);
                        float w = (q_wt - q_z) * scale;
                        sum = fmaf(w, h2f(input[ys + i]), sum);
                        packed_q_w >>= 4;
                    }
                }
            }
        }
        lane[tx] = sum;
    }
    if (g_order_variant) {
        float t = 0.f;
        for (int tx = 31; tx >= 0; tx--) t += lane[tx];
        return t;
    }
    return warp_sum32(lane);                                                  /* :205-207 */
}

/* plain fp64 dense dequant version of the same contraction (analytic cross-check) */
double orc_dot_q4_f64(int index, const f16 *input, const uint32_t *qw, const uint32_t *qz, const f16 *sc,
                      int inputElements, int pzh, int sh, int pwh) {
    double sum = 0.0;
    for (int k = 0; k < inputElements; k++) {
        uint32_t q = (qw[(size_t)index * pwh + k / 8] >> (4 * (k % 8))) & 0xF;
        int g = k / 128;
        uint32_t z = (qz[(size_t)index * pzh + g / 8] >> (4 * (g % 8))) & 0xF;
        double s = (double)h2f(sc[(size_t)index * sh + g]);
        sum += ((double)q - (double)z) * s * (double)h2f(input[k]);
    }
    return sum;
}

/* ---- a3: mat_vec_int4 / matmul(QWeight), gpu_kernels.h:213-240, llama2_q4.cu:224-233 */
/* output: base pointer; loff/pos as in the reference (loff=-1 => no kv addressing) */
void orc_matmul_q4(f16 *output, const f16 *input, const uint32_t *qw, const uint32_t *qz, const f16 *sc,
                   int inpSize, int opSize, int accum, int loff, int pos) {
    orc_init();
    int sh = divUp(inpSize, 128), pwh = divUp(inpSize, 32) * 4, pzh = divUp(sh, 8);
    f16 *out = output;
    if (loff != -1) out += (size_t)loff + (size_t)pos * opSize;               /* :225-227 */
#pragma omp parallel for schedule(static)
    for (int n = 0; n < opSize; n++) {
        float sum = orc_dot_q4(n, input, qw, qz, sc, inpSize, pzh, sh, pwh);
        if (accum) sum += h2f(out[n]);                                        /* :229-230 */
        out[n] = f2h(sum);                                                    /* :231 */
    }
}

void orc_matmul_q4_f64(double *out, const f16 *input, const uint32_t *qw, const uint32_t *qz, const f16 *sc,
                       int inpSize, int opSize) {
    orc_init();
    int sh = divUp(inpSize, 128), pwh = divUp(inpSize, 32) * 4, pzh = divUp(sh, 8);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < opSize; n++) out[n] = orc_dot_q4_f64(n, input, qw, qz, sc, inpSize, pzh, sh, pwh);
}

/* ---- a5: ffn_matvec_silu_kernel, gpu_kernels.h:256-275 ----------------------- */
void orc_ffn_matvec_silu(f16 *output, const f16 *input,
                         const uint32_t *gw, const uint32_t *gz, const f16 *gs,
                         const uint32_t *uw, const uint32_t *uz, const f16 *us, int inpSize, int opSize) {
    orc_init();
    int sh = divUp(inpSize, 128), pwh = divUp(inpSize, 32) * 4, pzh = divUp(sh, 8);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < opSize; n++) {
        float g_val = orc_dot_q4(n, input, gw, gz, gs, inpSize, pzh, sh, pwh);
        float u_val = orc_dot_q4(n, input, uw, uz, us, inpSize, pzh, sh, pwh);
        float val = g_val;
        val *= 1.0f / (1.0f + expf(-val));                                    /* :271 */
        val *= u_val;                                                         /* :272 */
        output[n] = f2h(val);
    }
}

/* ---- a6: mat_vec_kernel (fp16 GEMV), gpu_kernels.h:109-139, llama2_q4.cu:214-222 */
void orc_matmul_f16(f16 *out, const f16 *x, const f16 *w, int n, int d, float alpha) {
    orc_init();
    int numSerialLoads = divUp(divUp(n, 32), 8);
#pragma omp parallel for schedule(static)
    for (int index = 0; index < d; index++) {
        float lane[32];
        for (int tx = 0; tx < 32; tx++) {
            float sum = 0.f;
            for (int i = 0; i < numSerialLoads; i++) {
                int j = (i * 32 + tx) * 8;
                if (j < n)
                    for (int el = 0; el < 8; el++)
                        sum = fmaf(h2f(w[(size_t)index * n + j + el]), h2f(x[j + el]), sum);  /* :127-128 */
            }
            lane[tx] = sum;
        }
        float sum = warp_sum32(lane);
        sum *= alpha;                                                         /* :135 */
        out[index] = f2h(sum);
    }
}

/* ---- a7: rmsnorm_kernel, gpu_kernels.h:72-105 -------------------------------- */
void orc_rmsnorm(f16 *o, const f16 *x, const f16 *weight, int size) {
    orc_init();
    int ept = divUp(size, 1024);
    float part[1024];
    for (int t = 0; t < 1024; t++) {
        float ss = 0.f;
        for (int i = 0; i < ept; i++) {
            int index = t + i * 1024;
            if (index < size) { float v = h2f(x[index]); ss = fmaf(v, v, ss); }
        }
        part[t] = ss;
    }
    float ss = block_sum1024(part);
    ss /= size;                                                               /* :88 */
    ss += 1e-5f;                                                              /* :89 */
    ss = 1.0f / sqrtf(ss);                                                    /* :90 */
    for (int index = 0; index < size; index++) {
        float val = h2f(x[index]);
        val *= ss * h2f(weight[index]);                                       /* :101 */
        o[index] = f2h(val);
    }
}

/* ---- a8: RoPERotation_kernel, gpu_kernels.h:332-355 -------------------------- */
/* sk: pointer to this position's key row (sk_base + loff + pos*kv_dim in the reference) */
void orc_rope(f16 *sq, f16 *sk, int num_heads, int num_kv_heads, int head_size, int pos, float rope_theta) {
    orc_init();
    for (int h = 0; h < num_heads; h++) {
        f16 *q = sq + (size_t)h * head_size;
        for (int i = 0; i < head_size / 2; i++) {
            int head_dim = (i * 2) % head_size;
            float freq = 1.0f / powf(rope_theta, head_dim / (float)head_size);   /* :339 */
            float val = pos * freq;
            float fcr = cosf(val), fci = sinf(val);
            float q0 = h2f(q[i]), q1 = h2f(q[i + head_size / 2]);
            q[i] = f2h(q0 * fcr - q1 * fci);                                     /* :345 (no fma: see note) */
            q[i + head_size / 2] = f2h(q0 * fci + q1 * fcr);
            if (h < num_kv_heads) {
                f16 *k = sk + (size_t)h * head_size;
                float k0 = h2f(k[i]), k1 = h2f(k[i + head_size / 2]);
                k[i] = f2h(k0 * fcr - k1 * fci);
                k[i + head_size / 2] = f2h(k0 * fci + k1 * fcr);
            }
        }
    }
}

/* ---- a9-a11: MultiHeadAttention, llama2_q4.cu:267-284 ------------------------
 * mat_vec_kernel_simple gpu_kernels.h:142-168, softmax_kernel :357-401 (max_seq_len <= MAX_SEQ_LEN_SMEM_KERNEL = 8192) or
 * softmax_kernel_no_smem :403-446 (above: llama2_q4.cu:276-279 -- max_seq_len is the graph's sequence-length bin), vec_mat_kernel
 * :279-329. kc/vc point at this layer's cache (key_cache + loff). att row stride is pos+1 (P4). */
#define ORC_MAX_SEQ_LEN_SMEM_KERNEL 8192                                        /* common.h */
void orc_attention_bin(f16 *out, const f16 *q, const f16 *kc, const f16 *vc, f16 *att,
                       int num_heads, int head_size, int kv_mul, int pos, int max_seq_len) {
    orc_init();
    int dim = head_size * num_heads;
    int kv_dim = dim / kv_mul;
    int size = pos + 1;
    float alpha = (float)(1.0 / sqrt((double)head_size));                     /* llama2_q4.cu:273 */
#pragma omp parallel for schedule(static)
    for (int h = 0; h < num_heads; h++) {
        const f16 *qh = q + (size_t)h * head_size;
        const f16 *kh = kc + (size_t)(h / kv_mul) * head_size;
        const f16 *vh = vc + (size_t)(h / kv_mul) * head_size;
        f16 *a = att + (size_t)h * size;
        /* scores: one warp per (t,h); lane strided by 32 over head dim */
        for (int t = 0; t < size; t++) {
            float lane[32];
            for (int tx = 0; tx < 32; tx++) {
                float s = 0.f;
                for (int i = 0; i < divUp(head_size, 32); i++) {
                    int j = i * 32 + tx;
                    if (j < head_size) s = fmaf(h2f(kh[(size_t)t * kv_dim + j]), h2f(qh[j]), s);
                }
                lane[tx] = s;
            }
            float s = warp_sum32(lane);
            s *= alpha;
            a[t] = f2h(s);
        }
        /* softmax (fp32 in smem, strided partial max/sum over 1024 threads; order is
           reduction-unspecified in cub, restated sequentially per thread then tree) */
        float *f = (float *)malloc(sizeof(float) * size);
        for (int t = 0; t < size; t++) f[t] = h2f(a[t]);
        float max_val = f[0];
        for (int t = 1; t < size; t++) if (f[t] > max_val) max_val = f[t];
        float part[1024];
        for (int tid = 0; tid < 1024; tid++) {
            float sum = 0.f;
            for (int i = tid; i < size; i += 1024) { f[i] = expf(f[i] - max_val); sum += f[i]; }
            part[tid] = sum;
        }
        float sum = block_sum1024(part);
        if (max_seq_len <= ORC_MAX_SEQ_LEN_SMEM_KERNEL) {
            for (int t = 0; t < size; t++) a[t] = f2h(f[t] / sum);           /* :400 */
        } else {
            /* softmax_kernel_no_smem: the exponential is stored to `arr` as fp16 (:432) while the fp32 value is summed (:433);
               the normalisation divides the ROUNDED exponential (:445) */
            for (int t = 0; t < size; t++) a[t] = f2h(h2f(f2h(f[t])) / sum);
        }
        free(f);
        /* att . V : lane = k index mod 32, sequential over 32-chunks, then warp sum (:304-325) */
        for (int n = 0; n < head_size; n++) {
            float lane[32];
            for (int tx = 0; tx < 32; tx++) {
                float s = 0.f;
                for (int k = tx; k < size; k += 32)
                    s = fmaf(h2f(vh[(size_t)k * kv_dim + n]), h2f(a[k]), s);
                lane[tx] = s;
            }
            out[(size_t)h * head_size + n] = f2h(warp_sum32(lane));
        }
    }
}

void orc_attention(f16 *out, const f16 *q, const f16 *kc, const f16 *vc, f16 *att,
                   int num_heads, int head_size, int kv_mul, int pos) {
    orc_attention_bin(out, q, kc, vc, att, num_heads, head_size, kv_mul, pos, 0);
}

/* the sequence-length bin run_transformer launches run_llama_network with, llama2_q4.cu:354-360 */
int orc_seq_len_bin(int pos, int model_seq_len) {
    int seq_len = pos + 1;
    int graphIndex, seq_len_bin = 128;
    for (graphIndex = 0; graphIndex < 8 - 1; seq_len_bin *= 2, graphIndex++)  /* MAX_GRAPHS = 8, llama2_q4.cu:342 */
        if (seq_len <= seq_len_bin) break;
    if ((seq_len > seq_len_bin) || (graphIndex == 8 - 1)) seq_len_bin = model_seq_len;
    return seq_len_bin;
}

/* ---- a13: argmax_kernel, gpu_kernels.h:448-493 (ties -> lowest index, a legal outcome) */
int orc_argmax(const f16 *x, int size) {
    orc_init();
    float best = h2f(x[0]);
    int pos = 0;
    for (int i = 1; i < size; i++) {
        float v = h2f(x[i]);
        if (v > best) { best = v; pos = i; }
    }
    return pos;
}

/* ---- perplexity.h:3-51 -------------------------------------------------------- */
static void softmax_f32(float *x, int size) {
    float max_val = x[0];
    for (int i = 1; i < size; i++) if (x[i] > max_val) max_val = x[i];
    float sum = 0.0f;
    for (int i = 0; i < size; i++) { x[i] = expf(x[i] - max_val); sum += x[i]; }
    for (int i = 0; i < size; i++) x[i] /= sum;
}

float orc_compute_perplexity(const int *tokens, float *logits, int num_tokens, int vocab_size) {
    double sum = 0.0;
    for (int i = 0; i < num_tokens; i++) {
        softmax_f32(&logits[(size_t)i * vocab_size], vocab_size);
        double prob = logits[(size_t)i * vocab_size + tokens[i]];
        sum += log(prob);
    }
    return (float)exp(-(sum / num_tokens));
}

/* ---- sampler.h:31-40 ------------------------------------------------------------ */
unsigned int orc_random_u32(unsigned long long *state) {
    *state ^= *state >> 12;
    *state ^= *state << 25;
    *state ^= *state >> 27;
    return (unsigned int)((*state * 0x2545F4914F6CDD1Dull) >> 32);
}
float orc_random_f32(unsigned long long *state) { return (orc_random_u32(state) >> 8) / 16777216.0f; }

/* balanced pairwise tree over n = 2^k values in natural order */
static float pairwise_sum(const float *v, int n) {
    if (n == 1) return v[0];
    return pairwise_sum(v, n / 2) + pairwise_sum(v + n / 2, n / 2);
}

/* ---- temperature / top-p sampling: sampler.h:51-81, gpu_kernels.h:499-584 ----------
 * softmax_logits_kernel (fp16 rounding at: logits/temperature, exp, normalise),
 * descending sort of fp16 probabilities (cub radix sort is stable: equal keys keep
 * ascending index order), fp16 inclusive prefix sum (order of cub DeviceScan is
 * unspecified; restated in the fixed thread/lane/wave order documented below),
 * first index whose prefix >= threshold (else n-1). Returns the token id. */
static int cmp_desc_stable(const void *a, const void *b) {
    const uint32_t *pa = (const uint32_t *)a, *pb = (const uint32_t *)b;
    /* key = fp16 bits of a non-negative probability: monotonic as unsigned */
    if (pa[0] != pb[0]) return pa[0] > pb[0] ? -1 : 1;
    return pa[1] < pb[1] ? -1 : (pa[1] > pb[1] ? 1 : 0);
}
int orc_sample_topp(f16 *logits, int n, float temperature, float topp, float coin) {
    orc_init();
    for (int t = 0; t < n; t++) { float v = h2f(logits[t]); v /= temperature; logits[t] = f2h(v); }
    float max_val = h2f(logits[0]);
    for (int i = 1; i < n; i++) if (h2f(logits[i]) > max_val) max_val = h2f(logits[i]);
    float part[1024];
    for (int tid = 0; tid < 1024; tid++) {
        float sum = 0.f;
        for (int i = tid; i < n; i += 1024) { float v = expf(h2f(logits[i]) - max_val); logits[i] = f2h(v); sum += v; }
        part[tid] = sum;
    }
    /* cub's BlockReduce order is unspecified; restated as the balanced pairwise tree the HIP kernel uses */
    float sum = pairwise_sum(part, 1024);
    for (int t = 0; t < n; t++) logits[t] = f2h(h2f(logits[t]) / sum);
    uint32_t *kv = (uint32_t *)malloc(sizeof(uint32_t) * 2 * n);
    for (int t = 0; t < n; t++) { kv[2 * t] = logits[t]; kv[2 * t + 1] = (uint32_t)t; }
    float threshold;
    if (topp <= 0 || topp >= 1) threshold = coin;
    else { qsort(kv, n, 8, cmp_desc_stable); threshold = coin * topp; }
    /* fp16 inclusive prefix sum in the HIP kernel's fixed, cub-shaped order (cub::DeviceScan's own order is
     * unspecified): 1024 "threads" each sum their E consecutive entries sequentially, a Hillis-Steele scan runs over the
     * 64 lanes of each of the 16 "waves", the wave totals are added sequentially; every add rounds to fp16 (sampler.h:72-78
     * scans `half`). Then the first index whose prefix >= threshold, else n-1 (gpu_kernels.h:560-577). */
    int E = (n + 1023) / 1024;
    float tot[1024], v[1024], wt[16];
    for (int t = 0; t < 1024; t++) {
        float r = 0.f;
        for (int i = t * E; i < n && i < t * E + E; i++) r = h2f(f2h(r + h2f((f16)kv[2 * i])));
        tot[t] = r;
        v[t] = r;
    }
    for (int off = 1; off < 64; off <<= 1) {
        float nv[1024];
        for (int t = 0; t < 1024; t++) nv[t] = (t % 64) >= off ? h2f(f2h(v[t] + v[t - off])) : v[t];
        memcpy(v, nv, sizeof(v));
    }
    for (int w = 0; w < 16; w++) wt[w] = v[w * 64 + 63];
    int min_index = n - 1;
    for (int t = 0; t < 1024 && min_index == n - 1; t++) {
        float base = 0.f;
        for (int w = 0; w < t / 64; w++) base = h2f(f2h(base + wt[w]));
        float excl = (t % 64) > 0 ? h2f(f2h(base + v[t - 1])) : base;
        float r = 0.f;
        for (int i = t * E; i < n && i < t * E + E; i++) {
            r = h2f(f2h(r + h2f((f16)kv[2 * i])));
            if (h2f(f2h(excl + r)) >= threshold) { min_index = i; break; }
        }
    }
    (void)tot;
    int tok = (int)kv[2 * min_index + 1];
    free(kv);
    return tok;
}

/* ---- a18: loader, llama2_q4.cu:157-202, 408-426 (host pointers into the file image) */
static const uint8_t *take(const uint8_t **p, size_t bytes) { const uint8_t *r = *p; *p += bytes; return r; }
static void take_q(OrcQWeight *w, const uint8_t **p, int height, int width) {
    size_t a, b, c;
    orc_qweight_sizes(height, width, &a, &b, &c);
    w->weight = (const uint32_t *)take(p, a * 4);                              /* :167 */
    w->zeros = (const uint32_t *)take(p, b * 4);                               /* :168 */
    w->scales = (const f16 *)take(p, c * 2);                                   /* :169 */
}

OrcModel *orc_load(const char *path) {
    orc_init();
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    size_t bytes = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    OrcModel *m = (OrcModel *)calloc(1, sizeof(OrcModel));
    m->blob = (uint8_t *)malloc(bytes);
    m->blob_bytes = bytes;
    if (fread(m->blob, 1, bytes, f) != bytes) { fclose(f); free(m->blob); free(m); return NULL; }
    fclose(f);
    memcpy(&m->cfg, m->blob, sizeof(OrcConfig));                                /* :414 */
    OrcConfig *p = &m->cfg;
    int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    const uint8_t *c = m->blob + sizeof(OrcConfig);
    m->embed = (const f16 *)take(&c, (size_t)p->vocab_size * p->dim * 2);       /* :180 */
    m->wcls = (const f16 *)take(&c, (size_t)p->vocab_size * p->dim * 2);        /* :181 */
    m->rms_final = (const f16 *)take(&c, (size_t)p->dim * 2);                   /* :182 */
    m->layers = (OrcLayer *)calloc(p->n_layers, sizeof(OrcLayer));
    for (int i = 0; i < p->n_layers; i++) {
        OrcLayer *L = &m->layers[i];
        take_q(&L->q, &c, p->dim, p->dim);                                      /* :186-189 */
        take_q(&L->k, &c, p->dim, kv_dim);
        take_q(&L->v, &c, p->dim, kv_dim);
        take_q(&L->o, &c, p->dim, p->dim);
        take_q(&L->up, &c, p->dim, p->hidden_dim);                              /* :191 up BEFORE gate (P5) */
        take_q(&L->gate, &c, p->dim, p->hidden_dim);                            /* :192 */
        take_q(&L->down, &c, p->hidden_dim, p->dim);                            /* :193 */
        L->rms_att = (const f16 *)take(&c, (size_t)p->dim * 2);                 /* :195 */
        L->rms_ffn = (const f16 *)take(&c, (size_t)p->dim * 2);                 /* :196 */
    }
    if ((size_t)(c - m->blob) != bytes) {
        fprintf(stderr, "orc_load: file size %zu != expected %zu\n", bytes, (size_t)(c - m->blob));
        free(m->layers); free(m->blob); free(m);
        return NULL;
    }
    /* run state, llama2_q4.cu:38-67 (att sized n_heads*seq_len, what is actually needed) */
    m->x = (f16 *)calloc(p->dim, 2);
    m->xb = (f16 *)calloc(p->dim, 2);
    m->hb = (f16 *)calloc(p->hidden_dim, 2);
    m->q = (f16 *)calloc(p->dim, 2);
    m->att = (f16 *)calloc((size_t)p->n_heads * p->seq_len, 2);
    m->logits = (f16 *)calloc(p->vocab_size, 2);
    m->key_cache = (f16 *)calloc((size_t)p->n_layers * p->seq_len * kv_dim, 2);
    m->value_cache = (f16 *)calloc((size_t)p->n_layers * p->seq_len * kv_dim, 2);
    m->tokens = (int *)calloc((size_t)p->seq_len + 1, sizeof(int));
    m->pos = 0;
    return m;
}

void orc_free(OrcModel *m) {
    if (!m) return;
    free(m->x); free(m->xb); free(m->hb); free(m->q); free(m->att); free(m->logits);
    free(m->key_cache); free(m->value_cache); free(m->tokens); free(m->layers); free(m->blob);
    free(m->k64); free(m->v64); free(m);
}

const OrcConfig *orc_config(const OrcModel *m) { return &m->cfg; }
f16 *orc_logits(OrcModel *m) { return m->logits; }
f16 *orc_x(OrcModel *m) { return m->x; }
f16 *orc_key_cache(OrcModel *m) { return m->key_cache; }
f16 *orc_value_cache(OrcModel *m) { return m->value_cache; }
void orc_reset(OrcModel *m) { m->pos = 0; }
void orc_set_layer_dump(OrcModel *m, f16 *dump16, double *dump64) { m->dump16 = dump16; m->dump64 = dump64; }

/* ---- a15: run_llama_network, llama2_q4.cu:286-340: one token at position pos --- */
void orc_forward(OrcModel *m, int token, int pos) {
    OrcConfig *p = &m->cfg;
    int dim = p->dim, hidden_dim = p->hidden_dim;
    int head_size = dim / p->n_heads;
    int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads;
    int kv_mul = p->n_heads / p->n_kv_heads;
    f16 *x = m->x;
    memcpy(x, m->embed + (size_t)token * dim, (size_t)dim * 2);               /* :294 copy_embedding */
    for (int l = 0; l < p->n_layers; l++) {
        OrcLayer *L = &m->layers[l];
        orc_rmsnorm(m->xb, x, L->rms_att, dim);                               /* :300 */
        int loff = l * p->seq_len * kv_dim;                                   /* :303 */
        orc_matmul_q4(m->q, m->xb, L->q.weight, L->q.zeros, L->q.scales, dim, dim, 0, -1, 0);        /* :307/310 */
        orc_matmul_q4(m->key_cache, m->xb, L->k.weight, L->k.zeros, L->k.scales, dim, kv_dim, 0, loff, pos);
        orc_matmul_q4(m->value_cache, m->xb, L->v.weight, L->v.zeros, L->v.scales, dim, kv_dim, 0, loff, pos);
        orc_rope(m->q, m->key_cache + loff + (size_t)pos * kv_dim, p->n_heads, p->n_kv_heads, head_size, pos, p->rope_theta); /* :317 */
        orc_attention_bin(m->xb, m->q, m->key_cache + loff, m->value_cache + loff, m->att, p->n_heads, head_size, kv_mul, pos,
                          orc_seq_len_bin(pos, p->seq_len));                  /* :320, bin from run_transformer :354-360 */
        orc_matmul_q4(m->x, m->xb, L->o.weight, L->o.zeros, L->o.scales, dim, dim, 1, -1, 0);        /* :323 */
        if (m->dump16) memcpy(m->dump16 + (size_t)(2 * l) * dim, x, (size_t)dim * 2);
        orc_rmsnorm(m->xb, x, L->rms_ffn, dim);                               /* :326 */
        orc_ffn_matvec_silu(m->hb, m->xb, L->gate.weight, L->gate.zeros, L->gate.scales,
                            L->up.weight, L->up.zeros, L->up.scales, dim, hidden_dim);              /* :329 */
        orc_matmul_q4(m->x, m->hb, L->down.weight, L->down.zeros, L->down.scales, hidden_dim, dim, 1, -1, 0); /* :332 */
        if (m->dump16) memcpy(m->dump16 + (size_t)(2 * l + 1) * dim, x, (size_t)dim * 2);
    }
    orc_rmsnorm(x, x, m->rms_final, dim);                                     /* :336 (in place) */
    orc_matmul_f16(m->logits, x, m->wcls, p->dim, p->vocab_size, 1.0f);       /* :339 */
}

/* ---- the same network function without any intermediate rounding (double everywhere, weights and embeddings as
 * stored): the yardstick both the reference-order restatement above and the GPU path are measured against where two
 * valid fp16 evaluations drift apart (deep random-weight models). Not a restatement of reference code: it follows
 * run_llama_network's dataflow (llama2_q4.cu:286-340) with exact arithmetic in place of each kernel. Positions must be
 * fed in order 0,1,2..; `cap` positions of double KV cache are kept. logits_out: vocab doubles. Returns 0 on success. */
static void matvec_q4_f64(double *out, const double *x, const OrcQWeight *w, int K, int N, int accum) {
    int sh = divUp(K, 128), pwh = divUp(K, 32) * 4, pzh = divUp(sh, 8);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        double sum = 0.0;
        for (int g = 0; g < sh; g++) {
            uint32_t z = (w->zeros[(size_t)n * pzh + g / 8] >> (4 * (g % 8))) & 0xF;
            double sc = (double)h2f(w->scales[(size_t)n * sh + g]);
            double part = 0.0;
            int k1 = (g + 1) * 128 < K ? (g + 1) * 128 : K;
            for (int k = g * 128; k < k1; k++) {
                uint32_t q = (w->weight[(size_t)n * pwh + k / 8] >> (4 * (k % 8))) & 0xF;
                part += ((double)q - (double)z) * x[k];
            }
            sum += part * sc;
        }
        out[n] = accum ? out[n] + sum : sum;
    }
}
static void rmsnorm_f64(double *o, const double *x, const f16 *w, int n) {
    double ss = 0.0;
    for (int i = 0; i < n; i++) ss += x[i] * x[i];
    ss = 1.0 / sqrt(ss / n + 1e-5);                                            /* gpu_kernels.h:98-99 */
    for (int i = 0; i < n; i++) o[i] = x[i] * ss * (double)h2f(w[i]);
}
int orc_forward_f64(OrcModel *m, int token, int pos, int cap, double *logits_out) {
    OrcConfig *p = &m->cfg;
    int dim = p->dim, hidden = p->hidden_dim, hs = dim / p->n_heads;
    int kv_dim = (p->dim * p->n_kv_heads) / p->n_heads, kv_mul = p->n_heads / p->n_kv_heads;
    if (cap < 1 || pos < 0 || pos >= cap) return -1;
    if (m->f64_cap != cap) {
        free(m->k64); free(m->v64);
        m->k64 = (double *)calloc((size_t)p->n_layers * cap * kv_dim, sizeof(double));
        m->v64 = (double *)calloc((size_t)p->n_layers * cap * kv_dim, sizeof(double));
        m->f64_cap = cap;
        if (!m->k64 || !m->v64) return -2;
    }
    double *x = (double *)malloc(sizeof(double) * dim), *xb = (double *)malloc(sizeof(double) * dim);
    double *q = (double *)malloc(sizeof(double) * dim), *hb = (double *)malloc(sizeof(double) * hidden);
    double *hb2 = (double *)malloc(sizeof(double) * hidden), *att = (double *)malloc(sizeof(double) * (pos + 1));
    for (int i = 0; i < dim; i++) x[i] = (double)h2f(m->embed[(size_t)token * dim + i]);
    for (int l = 0; l < p->n_layers; l++) {
        OrcLayer *L = &m->layers[l];
        double *kc = m->k64 + ((size_t)l * cap + pos) * kv_dim, *vc = m->v64 + ((size_t)l * cap + pos) * kv_dim;
        rmsnorm_f64(xb, x, L->rms_att, dim);
        matvec_q4_f64(q, xb, &L->q, dim, dim, 0);
        matvec_q4_f64(kc, xb, &L->k, dim, kv_dim, 0);
        matvec_q4_f64(vc, xb, &L->v, dim, kv_dim, 0);
        for (int h = 0; h < p->n_heads; h++)                                   /* RoPE, gpu_kernels.h:332-355 */
            for (int i = 0; i < hs / 2; i++) {
                double freq = 1.0 / pow((double)p->rope_theta, (double)(2 * i) / (double)hs);
                double c = cos(pos * freq), s = sin(pos * freq);
                double a = q[h * hs + i], b = q[h * hs + i + hs / 2];
                q[h * hs + i] = a * c - b * s;
                q[h * hs + i + hs / 2] = a * s + b * c;
                if (h < p->n_kv_heads) {
                    a = kc[h * hs + i]; b = kc[h * hs + i + hs / 2];
                    kc[h * hs + i] = a * c - b * s;
                    kc[h * hs + i + hs / 2] = a * s + b * c;
                }
            }
        for (int h = 0; h < p->n_heads; h++) {                                 /* MHA, llama2_q4.cu:267-284 */
            const double *kb = m->k64 + (size_t)l * cap * kv_dim + (size_t)(h / kv_mul) * hs;
            const double *vb = m->v64 + (size_t)l * cap * kv_dim + (size_t)(h / kv_mul) * hs;
            double mx = -1e300, sum = 0.0;
            for (int t = 0; t <= pos; t++) {
                double sdot = 0.0;
                for (int i = 0; i < hs; i++) sdot += q[h * hs + i] * kb[(size_t)t * kv_dim + i];
                att[t] = sdot / sqrt((double)hs);
                if (att[t] > mx) mx = att[t];
            }
            for (int t = 0; t <= pos; t++) { att[t] = exp(att[t] - mx); sum += att[t]; }
            for (int i = 0; i < hs; i++) {
                double o = 0.0;
                for (int t = 0; t <= pos; t++) o += att[t] / sum * vb[(size_t)t * kv_dim + i];
                xb[h * hs + i] = o;
            }
        }
        matvec_q4_f64(x, xb, &L->o, dim, dim, 1);
        if (m->dump64) memcpy(m->dump64 + (size_t)(2 * l) * dim, x, (size_t)dim * sizeof(double));
        rmsnorm_f64(xb, x, L->rms_ffn, dim);
        matvec_q4_f64(hb, xb, &L->gate, dim, hidden, 0);
        matvec_q4_f64(hb2, xb, &L->up, dim, hidden, 0);
        for (int i = 0; i < hidden; i++) hb[i] = hb[i] / (1.0 + exp(-hb[i])) * hb2[i];   /* gpu_kernels.h:271-272 */
        matvec_q4_f64(x, hb, &L->down, hidden, dim, 1);
        if (m->dump64) memcpy(m->dump64 + (size_t)(2 * l + 1) * dim, x, (size_t)dim * sizeof(double));
    }
    rmsnorm_f64(xb, x, m->rms_final, dim);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < p->vocab_size; v++) {
        double sdot = 0.0;
        for (int i = 0; i < dim; i++) sdot += (double)h2f(m->wcls[(size_t)v * dim + i]) * xb[i];
        logits_out[v] = sdot;
    }
    free(x); free(xb); free(q); free(hb); free(hb2); free(att);
    return 0;
}

/* greedy decode, generate() llama2_q4.cu:436-482 with run_transformer :346-395 and
 * sample() sampler.h:43-50: feeds prompt tokens one per step, then argmax tokens.
 * out_tokens[0..steps] receives the token ring (prompt echoed); if logits_out != NULL it
 * receives vocab fp32 logits per step (the perplexity path's logits_array, :377-382). */
int orc_generate_greedy(OrcModel *m, const int *prompt, int n_prompt, int steps, int *out_tokens, float *logits_out) {
    OrcConfig *p = &m->cfg;
    for (int i = 0; i < n_prompt; i++) m->tokens[i] = prompt[i];
    int pos = 0;
    while (pos < steps) {
        orc_forward(m, m->tokens[pos], pos);
        if (logits_out)
            for (int v = 0; v < p->vocab_size; v++) logits_out[(size_t)pos * p->vocab_size + v] = h2f(m->logits[v]);
        int gen_token = pos >= n_prompt - 1;
        int next = orc_argmax(m->logits, p->vocab_size);
        if (gen_token) m->tokens[pos + 1] = next;                             /* gpu_kernels.h:486-487 */
        if (pos > 0 && m->tokens[pos] == 2) break;                            /* :472-477: token of the PREVIOUS step */
        pos++;
    }
    for (int i = 0; i <= pos && i <= steps; i++) out_tokens[i] = m->tokens[i];
    return pos;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
