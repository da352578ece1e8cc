#!/usr/bin/env python3
"""Per-wave cycle stamps (s_memtime) of the attention role inside the attention -> o-proj launch at a given context, next to
the per-block wall-clock records of tools/lab/timeline_block.py. Profiling build.
tools/lab/timeline_attn.py [context] [model] [blocks per head: 0, 2 or 4 V slices]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
model = sys.argv[2] if len(sys.argv) > 2 else "7b"
vslice = int(sys.argv[3]) if len(sys.argv) > 3 else -1
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
if vslice >= 0:
    L.q4_set_gemv_early(10, vslice)
tr = api.Transformer(path)
tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], ctx)
L.q4_set_use_graphs(0)
heads, nw = tr.config.n_heads, 8
dbg = api.DevBuf(nbytes=4096 * 4 * 8 + heads * nw * 8 * 8)
names = ["entry", "position loaded", "scores done (q, K arrived)", "barrier 1", "exp + sum, barrier 2", "P.V + row shuffles", "published"]
for rep in range(3):
    L.q4_set_debug_buffer(dbg.ptr)
    tr.run_transformer(True)
    api.synchronize()
    L.q4_set_debug_buffer(None)
    raw = dbg.get(np.uint64)
    t = raw[4096 * 4: 4096 * 4 + heads * nw * 8].reshape(heads * nw, 8).astype(np.int64)
    blk = raw[: 4096 * 4].reshape(4096, 4).astype(np.int64)
    blk = blk[blk[:, 0] > 0]
    att = blk[(blk[:, 3] & 0xFF) == 1]
    print("rep %d (position %d): attention blocks %.2f us (median entry -> end, wall clock)" % (rep, tr.pos() - 1, np.median(att[:, 2] - att[:, 0]) / 100.0))
    for a, b in [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (0, 6)]:
        d = t[:, b] - t[:, a]
        print("   %-28s -> %-28s median %6d  p90 %6d  max %6d cycles" % (names[a], names[b], np.median(d), np.percentile(d, 90), d.max()))
tr.close()
