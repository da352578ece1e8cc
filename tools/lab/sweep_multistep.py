#!/usr/bin/env python3
"""-n 256 token rate over the number of greedy steps per graph replay, interleaved repeats in ONE process (box-to-box
variance is ~1 %). Profiling build."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
res = {}
for rep in range(3):
    for k in (1, 2, 4, 8, 16):
        L.q4_set_gemv_early(7, k)
        tr.generate_ids(prompt, 256)
        res.setdefault(k, []).append(max(tr.generate_ids(prompt, 256)[1] for _ in range(3)))
for k, v in res.items():
    print("steps per replay %2d: %s tok/s" % (k, " ".join("%.1f" % x for x in v)))
tr.close()
