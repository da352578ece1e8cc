#!/usr/bin/env python3
"""ms per token inside each sequence-length bin for the fusion levels, interleaved repeats in one process (profiling build)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
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
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
EDGES = [128, 256, 512, 1024, 2048]


def secs(n):
    return min(tr.generate_ids(prompt, n)[3] for _ in range(3))


res = {}
for rep in range(2):
    for fusion in levels:
        L.q4_set_fusion(fusion)
        tr.generate_ids(prompt, 2048)
        t = [secs(n) for n in EDGES]
        per = [1e3 * t[0] / (EDGES[0] - 1)] + [1e3 * (t[i] - t[i - 1]) / (EDGES[i] - EDGES[i - 1]) for i in range(1, len(EDGES))]
        res.setdefault(fusion, []).append(per)
for fusion, runs in res.items():
    best = [min(r[i] for r in runs) for i in range(len(EDGES))]
    print("fusion %d: " % fusion + "  ".join("bin%d %.4f" % (e, b) for e, b in zip(EDGES, best)) + "  ms/token", flush=True)
tr.close()
