#!/usr/bin/env python3
"""Average launch duration of every launch class inside the eager decode network (HIP events on the launches), per fusion level.
tools/lab/classes.py [model] [levels]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "7b"
levels = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 3]
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
for rep in range(2):
    for lvl in levels:
        L.q4_set_fusion(lvl)
        tr = api.Transformer(path)
        tr.generate_ids(prompt, 64)
        out = []
        tot = 0.0
        for cls, name in ((1, "qkv"), (2, "attention"), (4, "o-proj"), (6, "attention+o-proj"), (8, "gate/up"), (16, "down"), (17, "down+qkv"), (32, "norm+classifier"), (64, "embedding")):
            try:
                a, mn, mx, n = tr.bench_in_network(cls, 4)
            except api.Q4Error:
                continue
            out.append("%s %.2f us x %d" % (name, a, n // 4))
            tot += a * n / 4
        print("level %d: %s | sum per token %.1f us" % (lvl, "; ".join(out), tot), flush=True)
        tr.close()
