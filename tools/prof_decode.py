#!/usr/bin/env python3
"""Eager (no hipGraph) greedy decode for rocprofv3 passes over the product's own launch sequence:
tools/prof_decode.py <model> <ntok> [fusion] [knob=value ...]   (knobs: the profiling build's q4_set_gemv_early(knob, value))"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "7b"
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 256
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
knobs = [tuple(int(v) for v in a.split("=")) for a in sys.argv[4:]]
if knobs:
    api.use_profiling_build()
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
L.q4_set_use_graphs(2)     # eager launches with the graph path's bins: every attention form on the positions the graphs run it for
if len(sys.argv) > 3:
    L.q4_set_fusion(int(sys.argv[3]))
for k, v in knobs:
    L.q4_set_gemv_early(k, v)
tr = api.Transformer(path)
toks, tps, timed, secs = tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], ntok)
print("%s -n %d eager: %d tokens, %.1f tokens/s" % (model, ntok, timed, tps))
tr.close()
