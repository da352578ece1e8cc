#!/usr/bin/env python3
"""Is a GEMV faster when its weights were touched into the L2s just before?  graph us/launch for: touch, gemv, touch+gemv."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
L.q4_set_touch_blocks.argtypes = [C.c_int]
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
g = lambda k: min(tr.bench_kernel_graph(k, 32, 20) for _ in range(3))
print("o-proj: touch %.2f  gemv %.2f  touch+gemv %.2f us" % (g(10), g(4), g(11)))
for nb in (200, 400, 600, 800, 1000, 1376):
    L.q4_set_touch_blocks(nb)
    print("gate/up, %4d of 1376 blocks (%.1f MB) touched: touch %.2f  gemv %.2f  touch+gemv %.2f us" % (nb, nb * 34.1e-3, g(12), g(0), g(13)))
tr.close()
