#!/usr/bin/env python3
"""Fused QKV launch: waves per block (blocks per CU) -- per launch inside a hipGraph and tokens/s. Profiling build.
tools/lab/sweep_qkv_waves.py [model]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "13b"
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
dim = tr.config.dim
for rep in range(2):
    for waves in (4, 5, 6, 8):
        L.q4_set_gemv_tune(2, 4, waves)
        L.q4_set_gemv_early(2, 4)          # (resets the captured graphs)
        g = min(tr.bench_kernel_graph(3, 32, 20) for _ in range(3))
        tps = max(tr.generate_ids(prompt, 256)[1] for _ in range(3))
        print("QKV %d waves per block (%.2f blocks per CU): %.2f us per launch in a graph, -n 256 %.1f tokens/s" % (
            waves, 3 * dim / (4 * waves) / 256.0, g, tps), flush=True)
L.q4_set_gemv_tune(2, 4, 4)
tr.close()
