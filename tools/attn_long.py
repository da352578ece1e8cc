#!/usr/bin/env python3
"""Attention cost per layer at long context: in-graph us per launch of the attention step at several positions."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
for pos in (63, 127, 255, 511, 1023, 1535, 2047):
    api.check(L.q4_memcpy_h2d(tr.state.contents.pos, np.array([pos], dtype=np.int32).ctypes.data, 4))
    g = min(tr.bench_kernel_graph(6, 32, 10) for _ in range(3))
    kv = (pos + 1) * 2 * 4096 * 2
    print("pos %4d: attention %.2f us per layer, KV %.1f MB -> %.0f GB/s" % (pos, g, kv / 1e6, kv / g / 1e3), flush=True)
tr.close()
