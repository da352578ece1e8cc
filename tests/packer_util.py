"""Synthetic AWQ dumps (what convert_awq_to_bin.py leaves on disk) for the weight_packer tests and goldens."""
import json
import os

import numpy as np

CFG = {"hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 1, "num_attention_heads": 2,
       "num_key_value_heads": 2, "vocab_size": 64, "max_position_embeddings": 32, "rope_theta": 10000.0}
GROUP = 128
MATS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.up_proj", "mlp.gate_proj", "mlp.down_proj"]


def cdiv(a, b):
    return (a + b - 1) // b


def mat_shape(cfg, name):
    d, h = cfg["hidden_size"], cfg["intermediate_size"]
    kv = d * cfg["num_key_value_heads"] // cfg["num_attention_heads"]
    return {"self_attn.q_proj": (d, d), "self_attn.k_proj": (d, kv), "self_attn.v_proj": (d, kv), "self_attn.o_proj": (d, d),
            "mlp.up_proj": (d, h), "mlp.gate_proj": (d, h), "mlp.down_proj": (h, d)}[name]


def write_awq_dump(d, old_format, seed, cfg=None):
    """Writes d/config.json and d/dump/<tensor>.bin files; returns the config dict."""
    cfg = dict(cfg or CFG)
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(d, "dump"), exist_ok=True)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    dim, vocab = cfg["hidden_size"], cfg["vocab_size"]

    def dump(name, arr):
        open(os.path.join(d, "dump", name + ".bin"), "wb").write(np.ascontiguousarray(arr).tobytes())

    dump("model.embed_tokens.weight", rng.standard_normal((vocab, dim)).astype(np.float16))
    dump("lm_head.weight", rng.standard_normal((vocab, dim)).astype(np.float16))
    dump("model.norm.weight", rng.standard_normal(dim).astype(np.float16))
    for l in range(cfg["num_hidden_layers"]):
        base = "model.layers.%d" % l
        for name in MATS:
            K, N = mat_shape(cfg, name)
            G = cdiv(K, GROUP)
            if old_format:   # row-major [K][N/8], [G][N/8], [G][N]
                dump(base + "." + name + ".qweight", rng.integers(0, 2 ** 32, size=(K, N // 8), dtype=np.uint32))
                dump(base + "." + name + ".qzeros", rng.integers(0, 2 ** 32, size=(G, N // 8), dtype=np.uint32))
                dump(base + "." + name + ".scales", rng.uniform(0.002, 0.004, size=(G, N)).astype(np.float16))
            else:            # already column-major; scales padded to a multiple of 8 rows
                dump(base + "." + name + ".qweight", rng.integers(0, 2 ** 32, size=(N, cdiv(K, 8)), dtype=np.uint32))
                dump(base + "." + name + ".qzeros", rng.integers(0, 2 ** 32, size=(N, cdiv(G, 8)), dtype=np.uint32))
                dump(base + "." + name + ".scales", rng.uniform(0.002, 0.004, size=(N, cdiv(G, 8) * 8)).astype(np.float16))
        dump(base + ".input_layernorm.weight", rng.standard_normal(dim).astype(np.float16))
        dump(base + ".post_attention_layernorm.weight", rng.standard_normal(dim).astype(np.float16))
    return cfg


def mask_zero_padding(data, cfg):
    """Zero the padding nibbles of every `zeros` tensor in a packed .bin image (the reference leaves
    out-of-bounds garbage there). Returns the masked bytes."""
    buf = bytearray(data)
    dim, vocab = cfg["hidden_size"], cfg["vocab_size"]
    off = 32 + 2 * vocab * dim * 2 + dim * 2
    for _ in range(cfg["num_hidden_layers"]):
        for name in MATS:
            K, N = mat_shape(cfg, name)
            G = cdiv(K, GROUP)
            pwh, pzh = cdiv(K, 8), cdiv(G, 8)
            off += pwh * N * 4
            z = np.frombuffer(bytes(buf[off: off + pzh * N * 4]), dtype=np.uint32).reshape(N, pzh).copy()
            valid = G - (pzh - 1) * 8
            if valid < 8:
                z[:, pzh - 1] &= np.uint32((1 << (4 * valid)) - 1)
            buf[off: off + pzh * N * 4] = z.tobytes()
            off += pzh * N * 4 + G * N * 2
        off += 2 * dim * 2
    assert off == len(buf), (off, len(buf))
    return bytes(buf)


def synthetic_state_dict(seed):
    """A torch state-dict shaped like an AWQ dump: 27 tensors incl. one fp32 non-weight entry and a non-tensor."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(2):
        for name in ("self_attn.q_proj", "mlp.down_proj"):
            sd["model.layers.%d.%s.qweight" % (i, name)] = torch.randint(-2 ** 31, 2 ** 31 - 1, (64, 8), generator=g, dtype=torch.int32)
            sd["model.layers.%d.%s.qzeros" % (i, name)] = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, 8), generator=g, dtype=torch.int32)
            sd["model.layers.%d.%s.scales" % (i, name)] = torch.rand((1, 64), generator=g).half()
        for name in ("input_layernorm", "post_attention_layernorm"):
            sd["model.layers.%d.%s.weight" % (i, name)] = torch.rand(64, generator=g).half()
        for j in range(4):
            sd["model.layers.%d.extra%d" % (i, j)] = torch.rand(3, generator=g).half()
    sd["model.embed_tokens.weight"] = torch.rand((16, 64), generator=g).half()
    sd["lm_head.weight"] = torch.rand((16, 64), generator=g).half()
    sd["model.norm.weight"] = torch.rand(64, generator=g).half()
    sd["model.rotary.inv_freq"] = torch.rand(32, generator=g)          # fp32, not a weight
    sd["not_a_tensor"] = "metadata"
    return sd
