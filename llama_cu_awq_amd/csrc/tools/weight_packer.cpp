// weight_packer -- offline repacker: per-tensor AWQ dumps (convert_awq_to_bin.py output) -> llama2_q4 `.bin`.
// Same command line and same output bytes as the reference tool (weight_packer.cpp:233-297), written around a
// table of tensors instead of per-tensor code. Format notes:
//   header: 8 x 4 bytes (Config, common.h:9-18), parsed from HF config.json                 (ref :22-72, :256)
//   order : embed, lm_head, final norm; per layer q,k,v,o,UP,GATE,down, input_ln, post_ln    (ref :261-291)
//   QWeight on disk (column-major per output column n, K = input length):
//     weight[n][k/8] nibble k%8 = q[k][n];  zeros[n][g/8] nibble g%8 = z[g][n];  scales[n][g] (g = k/128)
//   OldAwqFormat=1: inputs are row-major [k][n/8] words whose nibble i holds column 8*(n/8) + order[i],
//   order = {0,2,4,6,1,3,5,7} (AWQ's interleave)                                              (ref :94-127)
//   OldAwqFormat=0: qweight/qzeros already in the output layout, scales padded to a multiple of 8 rows (ref :165-169, :200-211)
// Padding nibbles of `zeros` (groups >= K/128) are written as 0 here; the reference leaves whatever its
// out-of-bounds read found -- they are never read by any kernel.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static const int kGroup = 128;

struct Header { int32_t dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len; float rope_theta; };

static int cdiv(int a, int b) { return (a + b - 1) / b; }

static bool find_number(const std::string& json, const char* key, double* out) {
    std::string pat = std::string("\"") + key + "\":";
    size_t p = json.find(pat);
    if (p == std::string::npos) return false;
    *out = atof(json.c_str() + p + pat.size());
    return true;
}

static bool parse_config(const std::string& json, Header* h) {
    struct { const char* key; int32_t* dst; bool required; } fields[] = {
        {"hidden_size", &h->dim, true}, {"intermediate_size", &h->hidden_dim, true}, {"num_hidden_layers", &h->n_layers, true},
        {"num_attention_heads", &h->n_heads, true}, {"vocab_size", &h->vocab_size, true}, {"max_position_embeddings", &h->seq_len, true}};
    double v;
    for (auto& f : fields) {
        if (!find_number(json, f.key, &v)) { printf("error parsing config.json %s not found", f.key); return false; }
        *f.dst = (int32_t)v;
    }
    h->n_kv_heads = find_number(json, "num_key_value_heads", &v) ? (int32_t)v : h->n_heads;   // ref :44-50
    h->rope_theta = find_number(json, "rope_theta", &v) ? (float)v : 10000.0f;                 // ref :64-70
    return true;
}

static bool read_file(const std::string& path, void* dst, size_t bytes) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { printf("\nUnable to open %s\n", path.c_str()); return false; }
    bool ok = fread(dst, 1, bytes, f) == bytes;
    fclose(f);
    if (!ok) printf("error reading weights from %s", path.c_str());
    return ok;
}

// row-major AWQ words [rows][cols/8] -> column-major words [cols][ceil(rows/8)], little-endian nibbles
static std::vector<uint32_t> to_column_major(const std::vector<uint32_t>& in, int rows, int cols) {
    static const int lane_of_nibble[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    const int out_h = cdiv(rows, 8), in_w = cdiv(cols, 8);
    std::vector<uint32_t> out((size_t)cols * out_h, 0u);
    for (int r = 0; r < rows; r++)
        for (int wx = 0; wx < in_w; wx++) {
            const uint32_t word = in[(size_t)r * in_w + wx];
            for (int i = 0; i < 8; i++) {
                const int c = wx * 8 + lane_of_nibble[i];
                if (c >= cols) continue;
                out[(size_t)c * out_h + r / 8] |= ((word >> (4 * i)) & 0xFu) << (4 * (r % 8));
            }
        }
    return out;
}

static bool pack_qweight(FILE* fp, const std::string& base, const char* name, int K, int N, bool old_format) {
    const int groups = cdiv(K, kGroup), wh = cdiv(K, 8), zh = cdiv(groups, 8);
    std::vector<uint32_t> qw, qz;
    std::vector<uint16_t> sc((size_t)groups * N);
    const std::string stem = base + "." + name;
    if (old_format) {
        const int in_w = cdiv(N, 8);
        std::vector<uint32_t> rw((size_t)K * in_w), rz((size_t)groups * in_w);
        std::vector<uint16_t> rs((size_t)groups * N);
        if (!read_file(stem + ".qweight.bin", rw.data(), rw.size() * 4)) return false;
        if (!read_file(stem + ".qzeros.bin", rz.data(), rz.size() * 4)) return false;
        if (!read_file(stem + ".scales.bin", rs.data(), rs.size() * 2)) return false;
        qw = to_column_major(rw, K, N);
        qz = to_column_major(rz, groups, N);
        for (int n = 0; n < N; n++)
            for (int g = 0; g < groups; g++) sc[(size_t)n * groups + g] = rs[(size_t)g * N + n];
    } else {
        const int padded = zh * 8;   // the AWQ repo pads scales to a multiple of 8 rows
        qw.resize((size_t)wh * N);
        qz.resize((size_t)zh * N);
        std::vector<uint16_t> rs((size_t)padded * N);
        if (!read_file(stem + ".qweight.bin", qw.data(), qw.size() * 4)) return false;
        if (!read_file(stem + ".qzeros.bin", qz.data(), qz.size() * 4)) return false;
        if (!read_file(stem + ".scales.bin", rs.data(), rs.size() * 2)) return false;
        for (int n = 0; n < N; n++)
            for (int g = 0; g < groups; g++) sc[(size_t)n * groups + g] = rs[(size_t)n * padded + g];
    }
    bool ok = fwrite(qw.data(), 4, qw.size(), fp) == qw.size() && fwrite(qz.data(), 4, qz.size(), fp) == qz.size() &&
              fwrite(sc.data(), 2, sc.size(), fp) == sc.size();
    if (!ok) printf("error writing output for %s", stem.c_str());
    return ok;
}

static bool copy_tensor(FILE* fp, const std::string& path, size_t bytes) {
    std::vector<uint8_t> buf(bytes);
    if (!read_file(path, buf.data(), bytes)) return false;
    if (fwrite(buf.data(), 1, bytes, fp) != bytes) { printf("error writing output file from input %s", path.c_str()); return false; }
    return true;
}

int main(int argc, char** argv) {
    if (argc != 5) {
        printf("usage: weight_packer <config.json from huggingface> <path_to_awq_bin_weights> <output_bin_filename> [OldAwqFormat: 0 or 1]\n");
        return 0;
    }
    const std::string dir = argv[2];
    const bool old_format = atoi(argv[4]) != 0;
    std::string json;
    {
        FILE* f = fopen(argv[1], "rb");
        if (!f) { printf("unable to open config file\n"); return 0; }
        char buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) json.append(buf, n);
        fclose(f);
        if (json.empty()) { printf("unable to read config file\n"); return 0; }
    }
    Header h;
    if (!parse_config(json, &h)) return 1;
    printf("\nModel params:- \ndim: %d \nhidden_dim: %d\nn_heads: %d\nn_kv_heads: %d\nn_layers: %d\nseq_len: %d\nvocab_size: %d\nrope_theta: %g\n",
           h.dim, h.hidden_dim, h.n_heads, h.n_kv_heads, h.n_layers, h.seq_len, h.vocab_size, h.rope_theta);
    FILE* fp = fopen(argv[3], "wb+");
    if (!fp) { printf("unable to open output file\n"); return 0; }
    if (fwrite(&h, sizeof(h), 1, fp) != 1) { printf("unable to write model metadata\n"); return 0; }
    const size_t emb = (size_t)h.vocab_size * h.dim * 2;
    if (!copy_tensor(fp, dir + "/model.embed_tokens.weight.bin", emb)) return 1;
    if (!copy_tensor(fp, dir + "/lm_head.weight.bin", emb)) return 1;
    if (!copy_tensor(fp, dir + "/model.norm.weight.bin", (size_t)h.dim * 2)) return 1;
    const int kv_dim = (h.dim * h.n_kv_heads) / h.n_heads;
    struct { const char* name; int K, N; } mats[] = {
        {"self_attn.q_proj", h.dim, h.dim}, {"self_attn.k_proj", h.dim, kv_dim}, {"self_attn.v_proj", h.dim, kv_dim},
        {"self_attn.o_proj", h.dim, h.dim}, {"mlp.up_proj", h.dim, h.hidden_dim}, {"mlp.gate_proj", h.dim, h.hidden_dim},
        {"mlp.down_proj", h.hidden_dim, h.dim}};
    for (int l = 0; l < h.n_layers; l++) {
        printf("\nProcessing weights for layer: %d\n", l);
        const std::string base = dir + "/model.layers." + std::to_string(l);
        for (auto& m : mats)
            if (!pack_qweight(fp, base, m.name, m.K, m.N, old_format)) return 1;
        if (!copy_tensor(fp, base + ".input_layernorm.weight.bin", (size_t)h.dim * 2)) return 1;
        if (!copy_tensor(fp, base + ".post_attention_layernorm.weight.bin", (size_t)h.dim * 2)) return 1;
    }
    printf("\nDone!\n");
    fclose(fp);
    return 0;
}
