#!/usr/bin/env python3
"""Run one timed kernel id in isolation (for rocprofv3 --pmc passes): tools/prof_kernel.py <kernel_id> [iters] [model] [knob value]
(a knob: the profiling build with q4_set_gemv_early(knob, value), e.g. 11 0 = the gate/up launch without the LDS-DMA engine)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

kid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = sys.argv[3] if len(sys.argv) > 3 else "7b"
knob = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else None
if knob:
    api.use_profiling_build()
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
if knob:
    L.q4_set_gemv_early(*knob)
avg, mn, mx = tr.bench_kernel(kid, iters)
print("kernel %d: avg %.2f us min %.2f max %.2f" % (kid, avg, mn, mx))
tr.close()
