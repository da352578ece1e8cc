#!/usr/bin/env python3
"""Per-bin cost of the fused attention-block launch vs the five-launch sequence (ms per token in bins 128 and 256)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

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
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]


def secs(n):
    tr.generate_ids(prompt, n)
    return min(tr.generate_ids(prompt, n)[3] for _ in range(3))


for fusion, early, att8 in ((1, 0, 0), (3, 0, 0), (2, 8 | (8 << 8), 0), (1, 0, 0), (3, 0, 0)):
    L.q4_set_fusion(fusion)
    L.q4_set_gemv_early(4, early)
    L.q4_set_gemv_early(6, att8)
    a, b, c = secs(128), secs(256), secs(512)
    print("fusion %d early %d att8 %d: bin128 %.4f ms/token, bin256 %.4f ms/token, bin512 %.4f ms/token" % (
        fusion, early & 255, att8, 1e3 * a / 127, 1e3 * (b - a) / 128, 1e3 * (c - b) / 256), flush=True)
tr.close()
