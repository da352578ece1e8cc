#!/usr/bin/env python3
"""Interleaved A/B of one profiling knob inside one process: greedy rings equal, tokens/s per value.
tools/lab/knob_ab.py <model> <ntok> <fusion> <knob> <v0,v1,...>"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth
api.use_profiling_build()
model, ntok, fusion, knob = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
vals = [int(v) for v in sys.argv[5].split(",")]
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
t = api.Transformer(path)
L.q4_set_fusion(fusion)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
L.q4_set_gemv_early(knob, vals[0])
want = t.generate_ids(prompt, ntok)[0].copy()
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        L.q4_set_gemv_early(knob, v)
        same = np.array_equal(t.generate_ids(prompt, ntok)[0], want)
        assert same or os.environ.get("RINGS_MAY_DIFFER"), v      # (a knob that changes an fp32 summation order: set RINGS_MAY_DIFFER=1)
        res[v].append(max(t.generate_ids(prompt, ntok)[1] for _ in range(3)))
for v in vals:
    print("%s -n %d level %d knob %d = %d: %s  median %.1f tokens/s" % (model, ntok, fusion, knob, v, " ".join("%.1f" % x for x in res[v]), float(np.median(res[v]))))
print("time-outs:", L.q4_handoff_timeouts(), "fusion:", L.q4_get_fusion())
t.close()
