#!/usr/bin/env python3
"""Wall-clock stamps (s_memrealtime, 10 ns) of the split-context attention -> o-proj launch at a given context: per wave of every
(head, chunk) block of the attention role -- entry, position known, q landed, scores done (the K rows have arrived), block maximum
known, P.V done (the V rows have arrived), record stored, head merged (chunk 0) -- and per block of the o-proj role (entry, wait
passed, end), all relative to the launch's first entry. Profiling build; the last layer's launch of one eager token.
tools/lab/timeline_split.py [context] [model] [ring: 0 registers, 1 LDS-DMA rings]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
model = sys.argv[2] if len(sys.argv) > 2 else "7b"
ring = int(sys.argv[3]) if len(sys.argv) > 3 else 1
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
L.q4_set_gemv_early(11, -1)     # wave-owned GEMVs: the strips kernels stamp into the same buffer
L.q4_set_gemv_early(14, ring)
tr = api.Transformer(path)
tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], ctx)
L.q4_set_use_graphs(0)
heads, nw = tr.config.n_heads, 8
NB = 4096
dbg = api.DevBuf(nbytes=NB * 4 * 8 + NB * nw * 8 * 8)
names = ["entry", "position known", "q landed", "scores done", "maximum known", "P.V done", "record stored", "head merged"]
pct = lambda v: "%5.2f / %5.2f / %5.2f / %5.2f" % tuple(np.percentile(v, [0, 50, 90, 100]))
for rep in range(3):
    dbg.put(np.zeros(NB * 4 + NB * nw * 8, np.uint64)) if hasattr(dbg, "put") else None
    L.q4_set_debug_buffer(dbg.ptr)
    tr.run_transformer(True)
    api.synchronize()
    L.q4_set_debug_buffer(None)
    raw = dbg.get(np.uint64)
    blk = raw[: NB * 4].reshape(NB, 4).astype(np.int64)
    blk = blk[blk[:, 0] > 0]
    t0 = blk[:, 0].min()
    att = blk[(blk[:, 3] & 0xFF) == 1]
    con = blk[(blk[:, 3] & 0xFF) == 2]
    w = raw[NB * 4:].reshape(NB * nw, 8).astype(np.int64)
    nblk = len(att)
    w = w[: nblk * nw]
    live = w[:, 3] > 0
    print("rep %d, position %d, ring %d: %d attention blocks (%d live waves), %d o-proj blocks; launch span %.2f us" % (
        rep, tr.pos() - 1, ring, nblk, live.sum(), len(con), (blk[:, 2].max() - t0) / 100.0))
    print("   (min / median / p90 / max, us since the launch's first entry)")
    for k in range(7):
        v = w[live][:, k]
        v = v[v > 0]
        if len(v):
            print("   %-16s %s" % (names[k], pct((v - t0) / 100.0)))
    mg = w[:, 7]
    mg = mg[mg > 0]
    if len(mg):
        print("   %-16s %s" % (names[7], pct((mg - t0) / 100.0)))
    d = w[live]
    print("   per wave: entry -> scores done %s" % pct((d[:, 3] - d[:, 0]) / 100.0))
    print("             maximum known -> P.V done %s" % pct((d[:, 5] - d[:, 4]) / 100.0))
    print("   attention blocks end %s" % pct((att[:, 2] - t0) / 100.0))
    if len(con):
        print("   o-proj blocks: entry %s" % pct((con[:, 0] - t0) / 100.0))
        print("                  wait passed %s" % pct((con[:, 1] - t0) / 100.0))
        print("                  end %s" % pct((con[:, 2] - t0) / 100.0))
tr.close()
