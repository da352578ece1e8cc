#!/usr/bin/env python3
"""Marginal cost of each launch class inside the token graph: run the -n 256 generation with one class of launches
left out (q4_set_skip_mask; outputs are garbage then) and print the ms/token difference.  tools/lab/breakdown.py [model]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only

model = sys.argv[1] if len(sys.argv) > 1 else "7b"
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


def run(mask):
    L.q4_set_skip_mask(mask)
    best = 1e9
    for _ in range(3):
        toks, tps, n, secs = tr.generate_ids(prompt, 256)
        if n > 50:
            best = min(best, 1e3 * secs / n)
    return best


NAMES = [(0, "full"), (1, "qkv"), (2, "attention"), (4, "o-proj"), (8, "gate/up"), (16, "down"), (32, "final norm + classifier"),
         (64, "embedding"), (1 | 2 | 4 | 8 | 16, "all layer kernels"), (127, "everything but argmax")]
full = None
for mask, name in NAMES:
    ms = run(mask)
    if full is None:
        full = ms
    nl = tr.config.n_layers if mask & 31 and not mask & 32 else 1
    print("%-28s %.4f ms/token   delta %.1f us  (%.2f us per layer-launch)" % (name, ms, 1e3 * (full - ms), 1e3 * (full - ms) / nl))
L.q4_set_skip_mask(0)
tr.close()
