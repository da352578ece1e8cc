// layer_attn.hip -- host side of the attention -> o-proj launch (layer_attn.h) and its head-128 instantiations.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <map>
#include "layer_attn.h"

namespace q4 {

int launch_attention_oproj_h128(int slots_kind, int att, dim3 grid, dim3 block, size_t smem, const AttOprojArgs& a, int* max_blocks_per_cu)
    Q4_AO_DISPATCH(16)

int launch_attention_oproj16(int att, dim3 grid, const AttOprojArgs& a, int* max_blocks_per_cu) {
    auto go = [&](auto kernel) -> int {
        if (max_blocks_per_cu) {
            int n = 0;
            Q4_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, LA16_WAVES * 64, LA16_LDS));
            *max_blocks_per_cu = n;
            return Q4_OK;
        }
        Q4_LAUNCH(kernel, grid, dim3(LA16_WAVES * 64), LA16_LDS, a.sync, a.att.pPos, a.att.q, a.att.key_cache, a.att.value_cache, a.oproj.m[0].w, a.oproj.m[0].z, a.oproj.m[0].s, a);
        Q4_LAUNCH_CHECK();
        return Q4_OK;
    };
    const bool k5120 = a.oproj.K == 5120;
    if (att == 5) return k5120 ? go(attention_oproj16_kernel<5, true>) : go(attention_oproj16_kernel<5, false>);
    if (att == 6) return k5120 ? go(attention_oproj16_kernel<6, true>) : go(attention_oproj16_kernel<6, false>);
    return Q4_ERR_UNSUPPORTED_SIZE;
}

static void fill_mat(GemvMat& m, const QWeight* w) { m.w = w->weight; m.z = w->zeros; m.s = w->scales; }

int g_ao_mute = 0;    // profiling build: q4_set_gemv_early(9, n): the attention blocks of the next n launches do not publish
int g_ao_vslice = 1;  // below the split-context bins: head_size / 32 attention blocks per head, one 64-byte V slice each (0: one block)
// split-context bins: the o-proj role requests its weights after this share of the K / V stream's duration (0: at entry). The duration is priced per context
// position from the slope q4_build_transformer measures with the stand-alone split-context attention launch (q4_runtime.hip measure_kv_price: 2.15 ns per
// position at Llama-2-7B on MI355X = 7.6 TB/s; the fused launch's own stream -- rows, records and then the held-back weights -- runs 1.3 x as long, which is
// why the optimum sits above 100 %). -1 = the measured optimum per bin (tools/lab/sweep_attn_hold.py, 7B, ms per token inside the bin at 0 / 130 / 180 / 230 %
// of that slope: bin 512 1.0933 / 1.0777 / 1.0710 / 1.0807, bin 1024 1.1229 / 1.1120 / 1.1255 / 1.1334, bin 2048 1.1925 / 1.1812 / 1.1979 / 1.2147)
int g_ao_hold_pct = -1;
static int ao_hold_pct(int seq_len_bin) { return g_ao_hold_pct >= 0 ? g_ao_hold_pct : seq_len_bin <= 512 ? 180 : 130; }
int g_ao_guard = 1;   // profiling build: q4_set_gemv_early(8, 0) admits grids larger than the resident capacity (forward-progress tests)

// CUs the launch stream may use: all of the device, or the bits of its CU mask (hipExtStreamCreateWithCUMask)
static std::map<hipStream_t, int>& stream_cu_cache() { static std::map<hipStream_t, int> c; return c; }
void attention_oproj_forget_stream(hipStream_t s) { stream_cu_cache().erase(s); }   // q4_stream_destroy: the handle may be reused
int stream_cu_count() {
    std::map<hipStream_t, int>& cache = stream_cu_cache();
    auto it = cache.find(g_stream);
    if (it != cache.end()) return it->second;
    int n = cu_count();
    if (g_stream) {
        uint32_t mask[16] = {};
        if (hipExtStreamGetCUMask(g_stream, 16, mask) == hipSuccess) {
            int bits = 0;
            for (int i = 0; i < 16; i++) bits += __builtin_popcount(mask[i]);
            if (bits > 0 && bits < n) n = bits;
        } else {
            (void)hipGetLastError();
        }
    }
    cache[g_stream] = n;
    return n;
}

// 10 ns ticks per context position of a model's K / V stream, by its hand-off words (q4_runtime.hip measures at build, forgets at free)
static std::map<const unsigned*, double>& kv_price() { static std::map<const unsigned*, double> m; return m; }
void attention_set_kv_price(const unsigned* sync, double ticks_per_pos) {
    if (ticks_per_pos > 0.0) kv_price()[sync] = ticks_per_pos; else kv_price().erase(sync);
}
double attention_get_kv_price(const unsigned* sync) { auto it = kv_price().find(sync); return it == kv_price().end() ? 0.0 : it->second; }

struct AoShape { int att, slots_kind, slots, nsp; AttOprojLaunch launch; };

int g_ao16 = 3;       // below the split-context bins: the launch on sixteen-wave blocks at dim 4096 (bit 0) / dim 5120 (bit 1) (profiling knob 20: 0 = the eight-wave launch)
// the sixteen-wave launch takes this case: V-slice forms, 128-wide heads, K = dim = 4096, four units per head, and its waiting blocks (dim / 16) leave a slot
static bool ao16_takes(const AoShape& s, int dim, int head_size, int n_heads) {
    const bool shape = (dim == 4096 && s.slots_kind == 0 && (g_ao16 & 1)) || (dim == 5120 && s.slots_kind == 1 && (g_ao16 & 2));    // K = dim in two k-slots, or two and the half slot
    if (!shape || (s.att != 5 && s.att != 6) || head_size != 128 || s.nsp != 4 || n_heads % 8) return false;
    static int per_cu = -1;
    if (per_cu < 0) { int n = 0; per_cu = launch_attention_oproj16(5, dim3(1), AttOprojArgs{}, &n) == Q4_OK ? n : 0; }
    return !g_ao_guard || (long long)per_cu * stream_cu_count() >= dim / 16 + 1;
}

// Which form the launch would use for this geometry and bin, or att = -1 when there is none (the caller then runs the
// stand-alone launches): head 64 / 128 / 256, multi-head or grouped-query, K = dim in <= 2 k-slots, 3 with a shared half slot,
// or 4; the one-block attention only while its score buffer fits the default 64 KB of LDS.
static AoShape ao_shape(int dim, int kv_dim, int head_size, int n_heads, int seq_len_bin, bool have_scratch, size_t scratch_bytes,
                        int split_min, int split_chunk) {
    AoShape s = {-1, 0, 0, 1, nullptr};
    const QGeom g = make_geom(dim, dim);
    if (g.nslots <= 2) { s.slots_kind = 0; s.slots = 2; }
    else if (g.nslots == 3 && g.pw4 - 2 * 64 <= 32) { s.slots_kind = 1; s.slots = 3; }
    else if (g.nslots == 4) { s.slots_kind = 2; s.slots = 4; }
    else return s;
    s.launch = head_size == 64 ? launch_attention_oproj_h64 : head_size == 128 ? launch_attention_oproj_h128 :
               head_size == 256 ? launch_attention_oproj_h256 : nullptr;
    if (!s.launch || !(kv_dim > 0 && dim % kv_dim == 0 && (dim % (LA_WAVES * la_ocols(s.slots))) == 0)) return s;
    // eight chunks per head = one attention block per CU at 32 heads (measured per bin, tools/lab/sweep_attn_bins.py), 64..256 positions
    const int auto_chunk = seq_len_bin <= 512 ? 64 : seq_len_bin <= 1024 ? 128 : 256;
    const int chunk = split_chunk ? split_chunk : auto_chunk;
    if (chunk != 64 && chunk != 128 && chunk != 256) return s;
    const int nsp = divUp(seq_len_bin, chunk);
    const bool split = seq_len_bin >= split_min && have_scratch && n_heads <= SYNC_MAX_HEADS &&
                       (size_t)n_heads * nsp * (head_size + ATT_REC_PAD) * sizeof(u32x2v) <= scratch_bytes;   // records as {float, tag} granules
    if (split) { s.att = chunk == 64 ? 4 : chunk == 128 ? 2 : 3; s.nsp = nsp; return s; }
    if ((size_t)(32 + LA_WAVES * head_size + seq_len_bin) * 4 > 64 * 1024) return s;
    s.att = seq_len_bin <= 128 ? 0 : 1;
    // (only where the bin is one register-resident group of the role, <= 256 positions: a model whose scratch does not hold
    // the split form's records runs the one-block form, which walks longer contexts group by group, above that too)
    if (g_ao_vslice && seq_len_bin <= 256) { s.att += 5; s.nsp = head_size / 32; }
    return s;
}

static size_t ao_smem(const AoShape& s, int head_size, int seq_len_bin) {
    const size_t smem_gemv = (size_t)s.slots * 256 * 16 + (size_t)s.slots * 512 + (size_t)s.slots * 256 * 4 + 16;
    const size_t smem_att = att_is_split(s.att) ? att_split_lds_bytes(LA_WAVES, head_size, 0) : (size_t)(32 + LA_WAVES * head_size + seq_len_bin) * 4;
    return smem_gemv > smem_att ? smem_gemv : smem_att;
}

int attention_oproj_form(int dim, int kv_dim, int head_size, int n_heads, int seq_len_bin, bool have_scratch, size_t scratch_bytes,
                         int split_min, int split_chunk) {
    const AoShape s = ao_shape(dim, kv_dim, head_size, n_heads, seq_len_bin, have_scratch, scratch_bytes, split_min, split_chunk);
    if (s.att < 0) return -1;
    const size_t smem = ao_smem(s, head_size, seq_len_bin);
    if (smem > AO_LDS_MAX) return -1;
    // Residency guard. Only the o-proj blocks wait, and only for attention blocks, which wait for nobody. Whatever order the
    // dispatcher picks, the launch cannot wedge as long as the waiting blocks alone cannot fill the stream's CUs: a slot is then
    // always left for an attention block, and every attention block that runs ends. (All blocks resident at once is the common
    // case; the split-context form of the last bin has more attention blocks than slots -- they queue behind each other.)
    const unsigned blocks = (unsigned)(n_heads * s.nsp + dim / (LA_WAVES * la_ocols(s.slots)));
    static std::map<unsigned long long, int> occupancy;     // per instantiation and LDS size (the query is a host call)
    const unsigned long long key = ((unsigned long long)head_size << 48) | ((unsigned long long)(s.slots_kind * 16 + s.att) << 32) | smem;
    auto it = occupancy.find(key);
    if (it == occupancy.end()) {
        int n = 0;
        if (s.launch(s.slots_kind, s.att, dim3(blocks), dim3(LA_WAVES * 64), smem, AttOprojArgs{}, &n) != Q4_OK) return -1;
        it = occupancy.insert({key, n}).first;
    }
    const int per_cu = it->second;
    // (split-context forms: the head's first chunk block waits for the other chunks' records -- one more waiter per head)
    const long long waiters = dim / (LA_WAVES * la_ocols(s.slots)) + (att_is_split(s.att) ? n_heads : 0);
    if (g_ao_guard && (long long)per_cu * stream_cu_count() < waiters + 1) return -1;
    return s.att;
}

int launch_attention_oproj(q4_half* x, q4_half* xb, const q4_half* q, const q4_half* key_cache, const q4_half* value_cache,
                           const QWeight* wo, int dim, int kv_dim, int n_heads, const int* pPos, int seq_len_bin, unsigned* sync,
                           float* scratch, size_t scratch_bytes, int split_min, int split_chunk) {
    const int head_size = dim / n_heads;
    const QGeom g = make_geom(dim, dim);
    const float alpha = (float)(1.0 / sqrt((double)head_size));                     // llama2_q4.cu:273
    const AoShape s = ao_shape(dim, kv_dim, head_size, n_heads, seq_len_bin, scratch != nullptr, scratch_bytes, split_min, split_chunk);
    if (s.att < 0) return Q4_ERR_UNSUPPORTED_SIZE;
    AttOprojArgs a = {};
    const int kv_mul = dim / kv_dim;
    a.att = {xb, q, key_cache, value_cache, head_size, kv_mul, kv_dim, pPos, alpha, seq_len_bin, nullptr};
    a.split = {scratch, q, key_cache, value_cache, head_size, kv_mul, kv_dim, pPos, alpha, xb, sync + SYNC_ARRIVE};
    GemvArgs& oa = a.oproj;
    oa.K = dim; oa.N = dim; oa.pw4 = g.pw4; oa.pzh = g.pzh; oa.sh = g.sh; oa.nslots = g.nslots;
    fill_mat(oa.m[0], wo);
    oa.out[0] = x; oa.x = xb; oa.accum = 1; oa.loff = -1;
    a.sync = sync;
    a.nheads = n_heads;
    a.natt = n_heads * s.nsp;
    a.no = dim / (LA_WAVES * la_ocols(s.slots));
    // ... where an o-proj block's share of the weights is small enough to arrive inside the ~4 us of hand-off hops behind the stream (7B, Mistral-7B:
    // 34 KB per block). Llama-2-13B's 160 blocks pull 85 KB each: held back they end the launch later (tools/lab/sweep_knob.py 13b 15 0,-1: bin 2048
    // 2.0285 -> 2.1179 ms per token), so they request at entry as before
    const size_t oproj_block_bytes = ((size_t)g.pw4 * 16 + (size_t)g.pzh * 4 + (size_t)g.sh * 2) * (size_t)dim / a.no;
    // ... and only on a stream that may use every CU: the price below was measured there (a CU-masked stream draws its rows at another rate)
    if (att_is_split(s.att) && ao_hold_pct(seq_len_bin) > 0 && (g_ao_hold_pct >= 0 || oproj_block_bytes <= 48 * 1024) && stream_cu_count() == cu_count()) {
        // time per context position of the K / V stream, in 10 ns ticks, Q16: measured for THIS model on THIS device when it was built
        // (q4_runtime.hip measure_kv_price); without a measurement (contexts below 1024) priced from the rows' bytes at the 7.6 TB/s measured on MI355X
        auto it = kv_price().find(sync);
        const double ticks_per_pos = it != kv_price().end() && it->second > 0.0 ? it->second : (4.0 * kv_dim) / 7.6e12 * 1e8;
        a.hold_q16 = (unsigned)(ticks_per_pos * 65536.0 * ao_hold_pct(seq_len_bin) / 100.0);
    }
#ifdef Q4_PROFILING
    a.att.dbg = g_dbg ? g_dbg + 4096 * 4 : nullptr;      // per-wave cycle stamps of the attention role, behind the per-block records
    a.split.dbg = a.att.dbg;                             // (split-context forms: wall-clock stamps, same place)
    a.dbg = g_dbg;
    if (g_ao_mute > 0) { a.mute = 1; g_ao_mute--; }
#endif
    if (ao16_takes(s, dim, head_size, n_heads)) {
        a.no = dim / 16;
        return launch_attention_oproj16(s.att, dim3(a.natt + a.no), a, nullptr);
    }
    const size_t smem = ao_smem(s, head_size, seq_len_bin);
    if (smem > AO_LDS_MAX) return Q4_ERR_UNSUPPORTED_SIZE;
    return s.launch(s.slots_kind, s.att, dim3(a.natt + a.no), dim3(LA_WAVES * 64), smem, a, nullptr);
}

// words of hand-off state a model needs: the error / epoch / arrival-counter page, then dim/2 granules of 8 bytes
size_t attention_sync_words(int dim) { return SYNC_GRANULES + line_area_words((size_t)dim / 2); }   // (whole lines one per 2 KiB: gemv_q4.h, line_slot)

}  // namespace q4
