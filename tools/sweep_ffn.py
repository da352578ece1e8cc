#!/usr/bin/env python3
"""gate/up launch-shape sweep: columns per wave x waves per block x early-bird waves per CU (in-graph us per launch)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
L.q4_set_gemv_early.argtypes = [C.c_int, C.c_int]
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
for cols, waves in ((2, 4),):
    for e in (0, 4, 8, 12, 16):
        for d in ((0,) if e == 0 else (7, 8, 9, 10)):
            L.q4_set_gemv_tune(3, cols, waves)
            L.q4_set_gemv_early(3, e | (d << 8))
            g = min(tr.bench_kernel_graph(0, 32, 20) for _ in range(3))
            print("gate/up cols %d waves %d early %2d delay %2d x128 cycles : %.2f us" % (cols, waves, e, d, g), flush=True)
L.q4_set_gemv_tune(3, 2, 4)
L.q4_set_gemv_early(3, 0)
for kid, kind, name in ((3, 2, "qkv"), (2, 1, "down")):
    for d in (0,):
        L.q4_set_gemv_early(kind, 4 | (d << 8))
        g = min(tr.bench_kernel_graph(kid, 32, 20) for _ in range(3))
        print("%s early 4 delay %2d x128 cycles : %.2f us" % (name, d, g), flush=True)
tr.close()
