#!/usr/bin/env python3
"""Per-wave s_memtime timeline of the attention kernel at a given context (profiling stamps via q4_set_debug_buffer)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth
api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
tr = api.Transformer(path)
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], ctx)      # fill the cache, pos = ctx
api.synchronize()
print("pos", tr.pos())
dbg = api.DevBuf(nbytes=32 * 16 * 8 * 8 + 4096)
L.q4_set_use_graphs(0)
print("attention graph-mode us/launch:", tr.bench_kernel_graph(6, 32, 20))
tr.bench_kernel(6, 40)
api.synchronize()
L.q4_set_debug_buffer(None)
t = dbg.get(np.uint64)[: 32 * 16 * 8].reshape(32 * 16, 8).astype(np.int64)
names = ["entry", "pos loaded", "scores done (K,V,q arrived)", "barrier 1", "barrier 2 (exp,sum)", "PV + shuffles done", "end"]
for a, b in [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (0, 6)]:
    d = t[:, b] - t[:, a]
    print("%-30s -> %-30s median %6d  p90 %6d  max %6d cycles" % (names[a], names[b], np.median(d), np.percentile(d, 90), d.max()))
tr.close()
