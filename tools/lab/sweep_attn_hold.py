#!/usr/bin/env python3
"""Split-context bins of the attention -> o-proj launch: ms per token inside bins 512 / 1024 / 2048 for (ring, hold) settings --
knob 14: the K / V rows in registers (0) or on LDS-DMA rings (1); knob 15: the o-proj role's weight requests held back this many per
cent of the K / V stream's estimated duration. Interleaved repeats in one process (profiling build).
tools/lab/sweep_attn_hold.py [model] [ring values] [hold values]   e.g. 7b 0,1 0,100,140,180"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
rings = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]   # (knob 14, the K / V rings of round 5, left the tree in round 6: the value is ignored)
holds = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,100,140,180").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
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
EDGES = [256, 512, 1024, 2048]


def secs(n):
    return min(tr.generate_ids(prompt, n)[3] for _ in range(3))


res, ring0 = {}, None
for rep in range(reps):
    for r in rings:
        for h in holds:
            L.q4_set_gemv_early(14, r)
            L.q4_set_gemv_early(15, h)
            toks = tr.generate_ids(prompt, EDGES[-1])[0]
            if ring0 is None:
                ring0 = toks.copy()
            same = len(toks) == len(ring0) and bool((toks == ring0).all())
            t = [secs(n) for n in EDGES]
            per = [1e3 * (t[i] - t[i - 1]) / (EDGES[i] - EDGES[i - 1]) for i in range(1, len(EDGES))]
            res.setdefault((r, h), []).append(per + [(EDGES[-1] - 1) / t[-1], same])
for (r, h), runs in res.items():
    best = [min(x[i] for x in runs) for i in range(3)]
    print("ring %d hold %3d%%: bin512 %.4f  bin1024 %.4f  bin2048 %.4f ms/token;  -n 2048 %.1f tok/s; tokens equal: %s" % (
        r, h, best[0], best[1], best[2], max(x[3] for x in runs), all(x[4] for x in runs)), flush=True)
L.q4_set_gemv_early(14, 0)
L.q4_set_gemv_early(15, -1)
tr.close()
