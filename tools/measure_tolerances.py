#!/usr/bin/env python3
"""Observed worst-case deviations of the HIP path from the CPU restatement on the test models (the numbers the
tolerances written in tests/ are 3x of). Prints one JSON line per case; run on the GPU box."""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from llama_cu_awq_amd import api, synth   # noqa: E402
import oracle                              # noqa: E402
from conftest import f16_ulp_diff          # noqa: E402

L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
d = tempfile.mkdtemp()
for name in ("tiny", "tiny_gqa", "small", "longk_gqa", "micro"):
    p = os.path.join(d, name + ".bin")
    synth.write_model(p, name, seed=7)
    for fusion in (1, 0):
        L.q4_set_fusion(fusion)
        t = api.Transformer(p)
        m = oracle.Model(p)
        prompt = [1, 17, 100, 45, 9]
        t.reset(prompt)
        toks = list(prompt)
        worst, worst_abs, kvw = 0.0, 0.0, 0.0
        steps = min(24, t.config.seq_len - 1)
        for pos in range(steps):
            gen = pos >= len(prompt) - 1
            t.run_transformer(gen)
            api.synchronize()
            ref = m.forward(toks[pos], pos).astype(np.float64)
            got = t.logits().astype(np.float64)
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))))
            worst_abs = max(worst_abs, float(np.max(np.abs(got - ref))))
            if gen:
                toks.append(int(np.argmax(ref)))
                ring = np.ctypeslib.as_array((C.c_int * t.config.seq_len).from_address(t.state.contents.shared_data + 4))
                ring[pos + 1] = toks[-1]
        rk, rv = m.kv()
        for layer in range(t.config.n_layers):
            for pos in (0, steps - 1):
                gk, gv = t.kv_row(layer, pos)
                kvw = max(kvw, float(np.max(np.abs(gk.astype(np.float64) - rk[layer, pos]) / np.maximum(1.0, np.abs(rk[layer, pos].astype(np.float64))))),
                          float(np.max(np.abs(gv.astype(np.float64) - rv[layer, pos]) / np.maximum(1.0, np.abs(rv[layer, pos].astype(np.float64))))))
        print(json.dumps({"model": name, "fusion": fusion, "logits_max_rel": worst, "logits_max_abs": worst_abs, "kv_max_rel": kvw, "max_logit": float(np.abs(ref).max())}), flush=True)
        t.close()
        m.close()
L.q4_set_fusion(1)

rng = np.random.default_rng(1234)
for (heads, kv_mul, hs, pos, seq, scratch) in [(4, 1, 64, 0, 64, 0), (32, 1, 128, 255, 256, 0), (32, 1, 128, 2047, 2048, 0), (8, 2, 128, 77, 128, 0), (8, 1, 32, 20, 128, 0),
                                               (32, 1, 128, 2047, 4096, 1), (32, 1, 128, 1000, 4096, 1), (8, 2, 128, 16000, 16384, 1), (8, 2, 128, 16000, 16384, 0),
                                               (32, 4, 64, 1500, 2048, 1), (32, 1, 64, 2047, 2048, 1)]:
    dim, kv_dim = heads * hs, heads * hs // kv_mul
    q = rng.standard_normal(dim).astype(np.float16)
    kc = rng.standard_normal(seq * kv_dim).astype(np.float16)
    vc = rng.standard_normal(seq * kv_dim).astype(np.float16)
    ref, _ = oracle.attention(q, kc, vc, heads, hs, kv_mul, pos)
    dq, dk, dv, do = api.DevBuf(q), api.DevBuf(kc), api.DevBuf(vc), api.DevBuf(nbytes=dim * 2)
    dpos = api.DevBuf(np.array([pos], dtype=np.int32))
    att = api.DevBuf(nbytes=heads * max(seq, dim) * 2 * 2) if scratch else None
    api.check(L.q4_multi_head_attention(do.ptr, dq.ptr, dk.ptr, dv.ptr, att.ptr if att else None, heads, hs, kv_mul, seq, dpos.ptr))
    api.synchronize()
    got = do.get(np.float16, dim)
    dd = f16_ulp_diff(got, ref)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    print(json.dumps({"attention": [heads, kv_mul, hs, pos, seq, scratch], "max_ulp": int(dd.max()), "frac_gt1": float((dd > 1).mean()), "frac_gt0": float((dd > 0).mean()),
                      "max_abs": float(err.max()), "max_ref": float(np.abs(ref.astype(np.float64)).max())}), flush=True)
