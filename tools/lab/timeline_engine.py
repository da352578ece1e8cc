#!/usr/bin/env python3
"""Wall-clock timeline (s_memrealtime, 10 ns ticks) of the loader / consumer gate/up engine (csrc/exp/ffn_engine.h), per block:
when the loader knows each fill landed, when consumer wave 0 sees and finishes each slot.  tools/lab/timeline_engine.py [lag]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth
api.use_profiling_build()
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
tr = api.Transformer(path)
lag = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = 256
dbg = api.DevBuf(nbytes=nb * 64 * 8 + 4096)
L.q4_set_gemv_early(11, lag)
L.q4_set_debug_buffer(dbg.ptr)
tr.bench_kernel(0, 40)          # several launches over the ring of layers, the buffer keeps the last one
api.synchronize()
t = dbg.get(np.uint64)[: nb * 64].reshape(nb, 64).astype(np.int64)
L.q4_set_debug_buffer(None)
t0 = min(t[:, 0].min(), t[:, 16].min())
us = lambda v: (v - t0) * 0.01


def row(name, col, sel=None):
    v = us(t[:, col] if sel is None else t[sel, col])
    v = v[v > -1e3]
    print("%-34s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))


nq = (t[:, 2:13] > 0).sum(axis=1)
print("engine setting %d; %d blocks, quads per block: %s" % (lag, nb, dict(zip(*np.unique(nq, return_counts=True)))))
row("loader entry", 0)
row("side data issued", 1)
for j in range(11):
    if (t[:, 2 + j] > 0).any():
        row("fill %d known landed" % j, 2 + j, t[:, 2 + j] > 0)
row("loader done", 15)
row("consumer entry", 16)
row("x staged", 17)
for i in range(11):
    sel = t[:, 18 + i] > 0
    if sel.any():
        row("slot %d seen landed (wave 0)" % i, 18 + i, sel)
        row("slot %d multiplied (wave 0)" % i, 31 + i, sel)
row("totals exchanged", 44)
row("outputs stored", 45)
d = np.diff(t[:, 2:2 + int(nq.min())], axis=1) * 0.01
print("fill-to-fill cadence per block (us): median %.3f p10 %.3f p90 %.3f  -> %.1f GB/s per CU, %.2f TB/s over %d CUs" % (
    np.median(d), np.percentile(d, 10), np.percentile(d, 90), 16384 / np.median(d) / 1e3, 16384 / np.median(d) / 1e6 * nb, nb))
b = int(np.argmax(t[:, 45]))
print("last block to finish: %d (quads %d); its rows:" % (b, nq[b]))
print("  loader:", " ".join("%.2f" % x for x in us(t[b, 0:16]) if x > -1e3))
print("  seen:  ", " ".join("%.2f" % x for x in us(t[b, 16:31]) if x > -1e3))
print("  done:  ", " ".join("%.2f" % x for x in us(t[b, 31:46]) if x > -1e3))
tr.close()
