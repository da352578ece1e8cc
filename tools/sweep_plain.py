#!/usr/bin/env python3
"""Plain int4 GEMV launch-shape sweep (BASELINE config 1 shapes): cols per wave x waves per block x early birds x K split."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
for kid, kind, name in ((1, 0, "4096->11008"), (2, 1, "11008->4096 accum")):
    for ks in (1, 0):
        L.q4_set_ksplit(ks)
        for cols in (4, 8):
            for waves in (4, 8):
                for e in (0, waves):
                    L.q4_set_gemv_tune(kind, cols, waves)
                    L.q4_set_gemv_early(kind, e)
                    try:
                        g = min(tr.bench_kernel_graph(kid, 32, 20) for _ in range(3))
                        a, mn, mx = tr.bench_kernel(kid, 128)
                        print("%-18s ksplit %d cols %d waves %d early %d : graph %.2f us, timestamps %.2f" % (name, ks, cols, waves, e, g, a), flush=True)
                    except Exception as ex:
                        print("%-18s ksplit %d cols %d waves %d early %d : failed %s" % (name, ks, cols, waves, e, ex))
tr.close()
