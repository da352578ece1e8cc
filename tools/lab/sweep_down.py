#!/usr/bin/env python3
"""Down projection (long-K plain int4 GEMV): K split in 2 / 3 balanced parts x block width, interleaved in one process
(profiling build). Per variant: the GEMV per call inside a hipGraph over the ring of layers, tokens/s of a -n 256 generation,
and the worst fp16-ulp distance from the CPU restatement on random inputs. tools/lab/sweep_down.py [model]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402
import oracle                              # noqa: E402  (checker only)

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
cfg = tr.config
K, N = cfg.hidden_dim, cfg.dim
rng = np.random.default_rng(5)
w = synth.random_qweight(rng, K, N)
x = rng.standard_normal(K).astype(np.float16)
ref = oracle.matmul_q4(x, *w, K, N)
dw, dx, dout = api.DevQWeight(*w), api.DevBuf(x), api.DevBuf(nbytes=N * 2)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]


def ulps(a, b):
    def key(v):
        u = np.ascontiguousarray(v, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
    return int(np.abs(key(a) - key(b)).max())


variants = [(2, 4), (4, 4)]
res = {v: [] for v in variants}
for rep in range(3):
    for ks, waves in variants:
        L.q4_set_ksplit(ks)
        L.q4_set_gemv_tune(1, 4, waves)
        L.q4_reset_graphs()
        api.matmul_q4(dout, dx, dw, K, N)
        api.synchronize()
        d = ulps(dout.get(np.float16, N), ref)
        tr.reset(prompt)
        g = min(tr.bench_kernel_graph(2, 32, 20) for _ in range(3))
        tps = max(tr.generate_ids(prompt, 256)[1] for _ in range(3))
        res[(ks, waves)].append((g, tps, d))
for (ks, waves), r in res.items():
    print("%s down %d->%d: K split %d, %2d waves/block: graph %.2f us per call, %.1f tokens/s (-n 256), max %d fp16 ulp vs restatement" % (
        model, K, N, ks, waves, min(v[0] for v in r), max(v[1] for v in r), max(v[2] for v in r)), flush=True)
L.q4_set_ksplit(2)
tr.close()
