#!/usr/bin/env python3
"""Wall-clock timeline (s_memrealtime, 10 ns ticks) of the "strips" form of the gate/up GEMV (csrc/gemv_strip.h; variants csrc/exp/ffn_strip_variants.h, knob 11 = 8..14),
per block: x chain of wave 0, when every wave's first piece landed and when it finished.  tools/lab/timeline_strip.py [setting]"""
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
setting = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nb = 256
dbg = api.DevBuf(nbytes=nb * 64 * 8 + 4096)
L.q4_set_gemv_early(11, setting)
L.q4_set_debug_buffer(dbg.ptr)
tr.bench_kernel(0, 40)          # several launches over the ring of layers, the buffer keeps the last one
api.synchronize()
t = dbg.get(np.uint64)[: nb * 64].reshape(nb, 64).astype(np.int64)
L.q4_set_debug_buffer(None)
t0 = t[:, 16:32].min()
us = lambda v: (v - t0) * 0.01


def row(name, v):
    v = us(v).ravel()
    print("%-40s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))


print("strip setting %d, %d blocks" % (setting, nb))
row("wave entry (all waves)", t[:, 16:32])
row("wave 0: x landed", t[:, 1])
row("wave 0: sum of squares exchanged", t[:, 2])
row("wave 0: x staged (barrier passed)", t[:, 3])
row("first piece landed (all waves)", t[:, 32:48])
for w in (0, 5, 10, 15):
    row("  wave %d first piece landed" % w, t[:, 32 + w])
for i in range(6):
    sel = t[:, 4 + i] > 0
    if sel.any():
        row("wave 0: unit %d multiplied" % i, t[sel, 4 + i])
row("last unit multiplied (all waves)", t[:, 48:64])
for w in (0, 5, 10, 15):
    row("  wave %d last unit multiplied" % w, t[:, 48 + w])
row("block: last wave done", t[:, 48:64].max(axis=1))
row("wave 0: totals written", t[:, 12])
row("outputs stored", t[:, 13])
b = int(np.argmax(t[:, 13]))
print("last block to finish: %d; wave 0 row:" % b, " ".join("%.2f" % x for x in us(t[b, 0:14])))
print("  first piece per wave:", " ".join("%.2f" % x for x in us(t[b, 32:48])))
print("  done per wave:       ", " ".join("%.2f" % x for x in us(t[b, 48:64])))
end = us(t[:, 13])
print("outputs stored by XCD (block % 8): " + " ".join("%.2f" % end[x::8].mean() for x in range(8)))
print("slowest 12 blocks: " + " ".join("%d:%.2f" % (b, end[b]) for b in np.argsort(-end)[:12]))
print("fastest 12 blocks: " + " ".join("%d:%.2f" % (b, end[b]) for b in np.argsort(end)[:12]))
# a second launch: are the same blocks late?
L.q4_set_gemv_early(11, setting)
L.q4_set_debug_buffer(dbg.ptr)
tr.bench_kernel(0, 33)
api.synchronize()
t2 = dbg.get(np.uint64)[: nb * 64].reshape(nb, 64).astype(np.int64)
L.q4_set_debug_buffer(None)
end2 = (t2[:, 13] - t2[:, 16:32].min()) * 0.01
print("second launch (another layer's weights): outputs stored median %.2f max %.2f; correlation of the blocks' end times with the first: %.2f" % (
    np.median(end2), end2.max(), np.corrcoef(end, end2)[0, 1]))
print("  by XCD: " + " ".join("%.2f" % end2[x::8].mean() for x in range(8)))
tr.close()
