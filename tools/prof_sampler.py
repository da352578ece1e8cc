#!/usr/bin/env python3
"""A short eager (no hipGraph) generation with the top-p sampler, for rocprofv3 --kernel-trace --stats."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
api.check(L.q4_set_device(0))
L.q4_set_use_graphs(0)
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
temp, topp = float(sys.argv[1]), float(sys.argv[2])
tr = api.Transformer(path, temperature=temp, topp=topp)
print(tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], 64)[1], "tokens/s")
tr.close()
