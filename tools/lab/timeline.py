#!/usr/bin/env python3
"""Per-wave s_memtime timeline of the fused gate/up kernel (ablation build 3): where do the cycles go?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth
api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
tr = api.Transformer(path)
nw = 11008 // 2
dbg = api.DevBuf(nbytes=nw * 8 * 8 + 4096)
L.q4_set_gemv_tune(3, 2, 4)
L.q4_set_gemv_early(3, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
L.q4_set_ablate(3)
L.q4_set_debug_buffer(dbg.ptr)
tr.bench_kernel(0, 40)          # several launches, buffer keeps the last one
api.synchronize()
t = dbg.get(np.uint64)[: nw * 8].reshape(nw, 8).astype(np.int64)
t0 = t[:, 0].min()
r = t - t0
names = ["entry", "loads issued", "x arrived+partials barrier", "staged (final barrier)", "slot0 done", "slot1 done", "(unused)", "end"]
print("cycles relative to the first wave's entry (s_memtime); %d waves" % nw)
for i, n in enumerate(names):
    if i == 6: continue
    c = r[:, i]
    print("%-28s min %7d  p10 %7d  median %7d  p90 %7d  max %7d" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
rt = (t[:, 6] - t[:, 6].min()) * 0.01           # us since the first wave's entry (constant 100 MHz clock)
life = (t[:, 7] - t[:, 0]) / 2.1e3               # us, shader clock ~2.1 GHz
print("entry time, us after the first wave: p10 %.2f median %.2f p90 %.2f max %.2f" % tuple(np.percentile(rt, [10, 50, 90, 100])))
print("end time,   us after the first wave: p10 %.2f median %.2f p90 %.2f max %.2f" % tuple(np.percentile(rt + life, [10, 50, 90, 100])))
h, _ = np.histogram(rt, bins=np.arange(0, rt.max() + 0.5, 0.5))
print("waves entering per 0.5 us:", h.tolist())
h, _ = np.histogram(rt + life, bins=np.arange(0, (rt + life).max() + 0.5, 0.5))
print("waves ending per 0.5 us:  ", h.tolist())
d = t[:, 7] - t[:, 0]
print("wave lifetime: median %d max %d cycles" % (np.median(d), d.max()))
for a, b in [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 7)]:
    dd = t[:, b] - t[:, a]
    print("phase %-26s -> %-26s median %6d  p90 %6d" % (names[a], names[b], np.median(dd), np.percentile(dd, 90)))
L.q4_set_ablate(0)
tr.close()
