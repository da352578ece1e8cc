#!/usr/bin/env python3
"""In-process A/B of one profiling knob (q4_set_gemv_early(kind, value)): ms per token inside each sequence-length bin,
interleaved repeats. tools/lab/sweep_knob.py model kind v0,v1[,v2...] [upto]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1]
kind = int(sys.argv[2])
values = [int(v) for v in sys.argv[3].split(",")]
upto = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
EDGES = [e for e in (128, 256, 512, 1024, 2048) if e <= upto]


def secs(n):
    return min(tr.generate_ids(prompt, n)[3] for _ in range(3))


res = {}
ring = {}
for rep in range(3):
    for v in values:
        L.q4_set_gemv_early(kind, v)
        toks = tr.generate_ids(prompt, EDGES[-1])[0]
        ring.setdefault(v, toks.copy())
        t = [secs(n) for n in EDGES]
        per = [1e3 * t[0] / (EDGES[0] - 1)] + [1e3 * (t[i] - t[i - 1]) / (EDGES[i] - EDGES[i - 1]) for i in range(1, len(EDGES))]
        res.setdefault(v, []).append(per + [(EDGES[-1] - 1) / t[-1], 255 / t[1] if len(t) > 1 else 0])
for v, runs in res.items():
    best = [min(r[i] for r in runs) for i in range(len(EDGES))]
    same = all((ring[v][:len(ring[values[0]])] == ring[values[0]]).all() for _ in (0,)) if len(ring[v]) == len(ring[values[0]]) else False
    print("knob %d = %d: " % (kind, v) + "  ".join("bin%d %.4f" % (e, b) for e, b in zip(EDGES, best)) +
          "  ms/token;  -n %d %.1f tok/s, -n 256 %.1f tok/s; token ring equal to first variant: %s" % (
              EDGES[-1], max(r[-2] for r in runs), max(r[-1] for r in runs), same), flush=True)
tr.close()
