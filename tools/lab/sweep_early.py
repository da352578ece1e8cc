#!/usr/bin/env python3
"""Early-bird sweep: the first E waves placed on each CU issue their weight loads D x 128 cycles after their x loads,
before the x staging completes (in-graph us per launch).  tools/lab/sweep_early.py [7b|13b]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only

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
big = tr.config.dim > 4096
for kid, kind, name in ((0, 3, "gate/up"), (3, 2, "qkv"), (2, 1, "down"), (4, 1 if big else 0, "o-proj")):
    for e, d in ((0, 0), (4, 0), (4, 2), (4, 4), (4, 6), (4, 8), (4, 10), (4, 12), (4, 16), (8, 0), (8, 8), (8, 12)):
        L.q4_set_gemv_early(kind, e | (d << 8))
        g = min(tr.bench_kernel_graph(kid, 32, 20) for _ in range(3))
        print("%-8s early %d delay %2d : %.2f us" % (name, e, d, g), flush=True)
    L.q4_set_gemv_early(kind, 0)
tr.close()
