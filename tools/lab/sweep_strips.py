#!/usr/bin/env python3
"""Where do strips (csrc/gemv_strip.h) overtake the wave-owned gate/up kernel? 8-layer bodies with dim 4096 and a hidden size of 40 .. 56
columns per CU: the norm-fused gate/up launch inside a hipGraph over the ring of layers, and tokens/s of the body, knob 11 = -1
(gemv_q4_kernel<MODE_FFN>) against 8 (strips, ring depth 2).  tools/lab/sweep_strips.py   (profiling build)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
dim = 4096
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
for cols in [int(c) for c in os.environ.get("COLS", "40,43,44,46,48,50,52,56").split(",")]:
    hidden = cols * 256
    path = "/tmp/llama2_q4_synth_d%d_h%d.bin" % (dim, hidden)
    geom = (dim, hidden, 8, dim // 128, dim // 128, 512, 256, 10000.0)
    if not os.path.exists(path):
        synth.write_model(path, geom)
    tr = api.Transformer(path)
    res = []
    for knob in (-1, 8, -1, 8):
        L.q4_set_gemv_early(11, knob)
        tr.generate_ids(prompt, 64)
        us = min(tr.bench_kernel_graph(0, 32, 20) for _ in range(3))
        tps = max(tr.generate_ids(prompt, 200)[1] for _ in range(3))
        res.append("%s %.2f us %.0f tok/s" % ("strips" if knob == 8 else "wave-owned", us, tps))
    print("hidden %5d (%d columns per CU): %s" % (hidden, cols, ";  ".join(res)), flush=True)
    tr.close()
    os.remove(path)
L.q4_set_gemv_early(11, 0)
