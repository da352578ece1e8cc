"""Seeded synthetic llama2_q4 checkpoints in the reference's exact `.bin` layout.

No real AWQ weights are available offline, so every bench/parity config runs on synthetic
weights in the real geometry.  Layout written here == what the reference
loader reads (llama2_q4.cu:172-202, sizes :82-98) == what weight_packer.cpp:256-291 writes:

    Config header (8 x 4 B)                                   common.h:9-18
    token_embedding_table [vocab, dim] fp16                   llama2_q4.cu:180
    wcls                  [vocab, dim] fp16                   :181
    rms_final_weight      [dim] fp16                          :182
    per layer: q, k, v, o, up, gate, down  (each: weight u32, zeros u32, scales fp16)   :186-193
               rms_att_weight [dim], rms_ffn_weight [dim]     :195-196

QWeight (common.h:20-24): column-major per output n; weight[n*K/8 + k/8] nibble k%8 (LSB first),
zeros[n*pzh + g/8] nibble g%8 (g = k/128), scales[n*G + g].

Distributions: weight/zeros nibbles iid uniform{0..15} (padding nibbles of `zeros` are random
garbage on purpose, like the reference packer's); scales U[0.002,0.004]; rms weights
1 +- 0.1; embedding N(0,1); wcls N(0,0.02) with row 2 (EOS) zeroed so greedy runs never stop early.
"""
import os
import struct

import numpy as np

GROUP_SIZE = 128

GEOMETRIES = {
    # name: (dim, hidden, layers, heads, kv_heads, vocab, seq_len, rope_theta)
    "7b": (4096, 11008, 32, 32, 32, 32000, 2048, 10000.0),
    "13b": (5120, 13824, 40, 40, 40, 32000, 2048, 10000.0),
    "mistral7b": (4096, 14336, 32, 32, 8, 32000, 2048, 1000000.0),   # GQA 4:1, the shape of Mistral-7B / CodeLlama-style models
    "tiny": (256, 352, 2, 4, 4, 512, 64, 10000.0),
    "tiny_gqa": (256, 352, 2, 8, 2, 512, 64, 10000.0),
    "small": (512, 1408, 3, 8, 8, 1024, 320, 10000.0),
    "micro": (64, 96, 2, 2, 2, 128, 32, 10000.0),       # committed fixture tests/golden/micro_model.bin (~60 KB)
    # Llama-2-70B / CodeLlama-34B traits at test size: GQA (kv_mul 4), head 128, down projection with K > 16384
    # (the K-split kernel's long form), rope_theta 1e6
    "longk_gqa": (1024, 20480, 2, 8, 2, 512, 128, 1000000.0),
    # the geometries that have a fused attention-block launch (multi-head, head 128): K = dim in two k-slots (7B-like) and
    # in three with the shared half slot (13B-like), contexts long enough to cross the 128 / 256 / 512 / 1024 bins
    "head128": (2560, 3584, 2, 20, 20, 512, 1100, 10000.0),
    "head128_k5120": (5120, 1408, 1, 40, 40, 512, 300, 10000.0),
    "head128_gqa": (2560, 3584, 2, 20, 4, 512, 700, 1000000.0),   # grouped-query (kv_mul 5) with a fused attention + o-proj form
    "head64_long": (512, 1408, 2, 8, 4, 512, 1300, 10000.0),     # head 64, grouped-query, context past the split threshold
    # seq_len 16384: the graph bins 4096 / 8192 and the last bin (= seq_len, llama2_q4.cu:360), where the reference's attention
    # switches to softmax_kernel_no_smem (> 8192, :276-279); head 64 and head 128, grouped-query
    "long16k": (512, 1408, 2, 8, 2, 512, 16384, 10000.0),
    "long16k_h128": (512, 1408, 2, 4, 2, 512, 16384, 1000000.0),
    # the other head sizes / K widths of the attention -> o-proj launch: TinyLlama-1.1B's shape (head 64, GQA 8:1, K = 2048 in
    # ONE k-slot), a head-256 model, and K = dim = 8192 in four k-slots with 70B-style GQA
    "tinyllama": (2048, 5632, 2, 32, 4, 512, 1100, 10000.0),
    "head256": (1024, 2816, 2, 4, 2, 512, 600, 10000.0),
    "head128_k8192": (8192, 1408, 1, 64, 8, 512, 300, 1000000.0),
    "head96": (768, 2048, 2, 8, 8, 512, 300, 10000.0),           # a head size that is no power of two (stand-alone attention, lanes masked)
    "head80_gqa": (640, 1728, 2, 8, 2, 512, 300, 10000.0),
    # the sampler at production vocabulary sizes on a one-layer body: 32000 = the register/LDS path with 32 keys per
    # thread, 40000 = the global-memory fallback (> 32 x 1024 entries)
    "v32k": (64, 96, 1, 2, 2, 32000, 32, 10000.0),
    "v40k": (64, 96, 1, 2, 2, 40000, 32, 10000.0),
    # the classifier's strips launch (dim 4096 / 5120, >= 64 vocabulary rows per CU) with the greedy sampler as its epilogue, on a
    # one-layer body: a vocabulary that splits evenly over 256 CUs and one that does not
    "cls4096": (4096, 1408, 1, 32, 32, 32000, 300, 10000.0),
    "cls5120": (5120, 1408, 1, 40, 40, 20000, 300, 10000.0),
    "cls4096_ragged": (4096, 1408, 1, 32, 32, 16600, 300, 10000.0),
    # the FFN half of a layer as one launch (csrc/gemv_ffn_pair.h, fusion level 4): Llama-2-7B's dim / hidden at two layers; a hidden size that
    # splits raggedly over 256 CUs (20.5 column pairs per CU; a last k-slot of 36 units); and the widest the launch's LDS holds (22 pairs, 48)
    "ffn_pair7b": (4096, 11008, 2, 32, 32, 512, 300, 10000.0),
    "ffn_pair_ragged": (4096, 10496, 2, 32, 32, 512, 300, 10000.0),
    "ffn_pair_wide": (4096, 11264, 2, 32, 32, 512, 300, 10000.0),
}


# named variants of a geometry: the same tensors from the same seeds with keyword changes of write_model.
#   7b_contractive: Llama-2-7B's shape with the scales of the matrices that WRITE the residual stream (o, down) x 0.25. A random-weight network of 32
#   layers is chaotic by construction -- two valid fp16 evaluations drift apart by 3..5e-2 of a logit (tests/test_baseline_configs_gpu.py) --; with the
#   residual branches damped a rounding difference is not amplified by the layers behind it, so SURVEY section 8(c)'s contract tolerance (3e-2) can be
#   exercised at full depth.
VARIANTS = {"7b_contractive": ("7b", {"residual_scale": 0.25})}


def geometry(name):
    return GEOMETRIES[VARIANTS[name][0]] if name in VARIANTS else GEOMETRIES[name]


def div_up(a, b):
    return (a - 1) // b + 1


def qweight_sizes(height, width):
    """(weight_u32, zeros_u32, scales_f16) element counts, llama2_q4.cu:82-98."""
    pwh = div_up(height, 32) * 4
    sh = div_up(height, GROUP_SIZE)
    pzh = div_up(sh, 8)
    return pwh * width, pzh * width, sh * width


def model_bytes(cfg):
    dim, hidden, layers, heads, kv_heads, vocab, _, _ = cfg
    kv_dim = dim * kv_heads // heads
    total = 32 + 2 * vocab * dim * 2 + dim * 2
    per = 0
    for (h, w) in ((dim, dim), (dim, kv_dim), (dim, kv_dim), (dim, dim), (dim, hidden), (dim, hidden), (hidden, dim)):
        a, b, c = qweight_sizes(h, w)
        per += a * 4 + b * 4 + c * 2
    per += 2 * dim * 2
    return total + layers * per


def _u32(rng, n):
    return np.frombuffer(rng.bytes(4 * n), dtype=np.uint32)


def _write_qweight(f, rng, height, width, scale_lo, scale_hi):
    a, b, c = qweight_sizes(height, width)
    f.write(_u32(rng, a).tobytes())
    f.write(_u32(rng, b).tobytes())
    f.write(rng.uniform(scale_lo, scale_hi, c).astype(np.float16).tobytes())


def write_model(path, cfg, seed=20240229, scale_lo=0.002, scale_hi=0.004, eos_row_zero=True, residual_scale=1.0):
    """Write a synthetic checkpoint; returns the byte size. One RNG stream per tensor group. residual_scale multiplies the scales of o and down."""
    if isinstance(cfg, str):
        if cfg in VARIANTS:
            base, kw = VARIANTS[cfg]
            return write_model(path, GEOMETRIES[base], seed=seed, scale_lo=scale_lo, scale_hi=scale_hi, eos_row_zero=eos_row_zero, **kw)
        cfg = GEOMETRIES[cfg]
    dim, hidden, layers, heads, kv_heads, vocab, seq_len, theta = cfg
    assert dim % 32 == 0 and hidden % 32 == 0, "packed height must be a multiple of 32 (SURVEY P6)"
    kv_dim = dim * kv_heads // heads
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(struct.pack("<7if", dim, hidden, layers, heads, kv_heads, vocab, seq_len, theta))
        rng = np.random.Generator(np.random.PCG64(seed))
        chunk = 1 << 22
        n = vocab * dim
        for off in range(0, n, chunk):  # embedding N(0,1)
            f.write(rng.standard_normal(min(chunk, n - off), dtype=np.float32).astype(np.float16).tobytes())
        wcls = (rng.standard_normal(n, dtype=np.float32) * 0.02).astype(np.float16).reshape(vocab, dim)
        if eos_row_zero and vocab > 2:
            wcls[2, :] = 0
        f.write(wcls.tobytes())
        del wcls
        f.write((1.0 + 0.1 * rng.uniform(-1, 1, dim)).astype(np.float16).tobytes())
        for l in range(layers):
            lr = np.random.Generator(np.random.PCG64([seed, l + 1]))
            _write_qweight(f, lr, dim, dim, scale_lo, scale_hi)      # q
            _write_qweight(f, lr, dim, kv_dim, scale_lo, scale_hi)   # k
            _write_qweight(f, lr, dim, kv_dim, scale_lo, scale_hi)   # v
            _write_qweight(f, lr, dim, dim, scale_lo * residual_scale, scale_hi * residual_scale)      # o
            _write_qweight(f, lr, dim, hidden, scale_lo, scale_hi)   # up   (before gate, P5)
            _write_qweight(f, lr, dim, hidden, scale_lo, scale_hi)   # gate
            _write_qweight(f, lr, hidden, dim, scale_lo * residual_scale, scale_hi * residual_scale)   # down
            f.write((1.0 + 0.1 * lr.uniform(-1, 1, dim)).astype(np.float16).tobytes())
            f.write((1.0 + 0.1 * lr.uniform(-1, 1, dim)).astype(np.float16).tobytes())
        size = f.tell()
    assert size == model_bytes(cfg), (size, model_bytes(cfg))
    os.replace(tmp, path)
    return size


def random_qweight(rng, height, width, scale_lo=0.002, scale_hi=0.004):
    """(weight u32[a], zeros u32[b], scales f16[c]) numpy arrays for kernel-level tests."""
    a, b, c = qweight_sizes(height, width)
    return (_u32(rng, a).copy(), _u32(rng, b).copy(), rng.uniform(scale_lo, scale_hi, c).astype(np.float16))


def dequant_dense(weight, zeros, scales, height, width):
    """fp64 dense [width, height] matrix from packed tensors (test helper, analytic check)."""
    pwh = div_up(height, 32) * 4
    sh = div_up(height, GROUP_SIZE)
    pzh = div_up(sh, 8)
    w = weight.reshape(width, pwh)
    k = np.arange(height)
    q = (w[:, k // 8] >> (4 * (k % 8)).astype(np.uint32)) & 0xF
    g = k // GROUP_SIZE
    z = (zeros.reshape(width, pzh)[:, g // 8] >> (4 * (g % 8)).astype(np.uint32)) & 0xF
    s = scales.reshape(width, sh)[:, g].astype(np.float64)
    return (q.astype(np.float64) - z.astype(np.float64)) * s


if __name__ == "__main__":
    import sys
    name, out = sys.argv[1], sys.argv[2]
    print(write_model(out, GEOMETRIES[name]), "bytes ->", out)
