#!/usr/bin/env python3
"""tokens/s of the -n 256 generation with the CLI's default sampler (temperature 0.5, top-p 0.6) and variants."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
for temp, topp in ((0.0, 0.9), (0.5, 0.6), (1.0, 0.9), (1.0, 1.0), (0.02, 0.9)):
    tr = api.Transformer(path, temperature=temp, topp=topp)
    tr.generate_ids(prompt, 256)
    best = max(tr.generate_ids(prompt, 256)[1] for _ in range(2))
    print("temperature %.2f top-p %.1f : %.1f tokens/s (%.4f ms/token)" % (temp, topp, best, 1e3 / best))
    tr.close()
