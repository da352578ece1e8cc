#!/usr/bin/env python3
"""Per-launch time of the fused q/k/v launch inside the eager network (HIP events on the product's own launches) with the wave-owned
kernel (profiling knob 11 = 0, the product) and as strips (8: every strips form wherever covered): tools/lab/qkv_strip_time.py [model]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "13b"
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
t = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
t.generate_ids(prompt, 64)
for rep in range(3):
    for knob in [int(v) for v in os.environ.get("KNOBS", "0,8").split(",")]:   # knob 11 values: csrc/exp/ffn_engine.h (0 the product, 8 every strips form wherever covered, 9 ring depth 4, 19 column units at K = 5120, -1 wave-owned only)
        L.q4_set_gemv_early(11, knob)
        t.reset(prompt)
        for pos in range(40):
            t.run_transformer(pos >= 7)
        api.synchronize()
        for mask, name in ((1, "qkv"), (8, "gate/up"), (16, "down")):
            avg, mn, mx, n = t.bench_in_network(mask, tokens=8)
            print("knob 11 = %2d  %-8s avg %.2f us  min %.2f  max %.2f  (%d launches)" % (knob, name, avg, mn, mx, n), flush=True)
t.close()
