#!/usr/bin/env python3
"""Sweep the int4 GEMV launch shapes (columns per wave x waves per block) and the ablation builds on a 7B-geometry
synthetic checkpoint; prints one line per configuration (pure kernel time from dispatch timestamps)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only
import bench                               # noqa: E402

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
kb = bench.kernel_bytes(tr.config)
KIND = {0: 3, 1: 0 if tr.config.dim <= 4096 else 1, 2: 1, 3: 2, 4: 0 if tr.config.dim <= 4096 else 1}   # kernel id -> tune slot


def run(kid, iters=192):
    """(graph steady-state us per launch, dispatch-timestamp avg us, GB/s from the graph number)"""
    tr.bench_kernel(kid, 32)
    avg, mn, mx = tr.bench_kernel(kid, iters)
    g = tr.bench_kernel_graph(kid, 32, 20)
    name, nbytes = kb[kid]
    return g, avg, nbytes / g / 1e3


for ks in (1, 0):
    L.q4_set_ksplit(ks)
    for kid in (1, 2, 4):
        for waves in (4, 8):
            L.q4_set_gemv_tune(KIND[kid], 4, waves)
            g, avg, gbs = run(kid)
            print("KSPLIT %d kernel %d %-30s waves %d : graph %7.2f us (timestamps %6.2f)  %7.1f GB/s" % (ks, kid, kb[kid][0], waves, g, avg, gbs), flush=True)
L.q4_set_ksplit(1)
COLSETS = {0: (2, 4), 1: (4, 8), 2: (4,), 3: (4,), 4: (4, 8)}
for kid in (0, 3):
    for cols in COLSETS[kid]:
        for waves in (4, 8):
            L.q4_set_gemv_tune(KIND[kid], cols, waves)
            try:
                avg, mn, gbs = run(kid)
                print("kernel %d %-32s cols %d waves %d : graph %7.2f us (timestamps %6.2f)  %7.1f GB/s" % (kid, kb[kid][0], cols, waves, avg, mn, gbs), flush=True)
            except Exception as e:
                print("kernel %d cols %d waves %d failed: %s" % (kid, cols, waves, e))
L.q4_set_gemv_tune(3, 2, 4)
for abl in (0, 1, 2, 4):
    L.q4_set_ablate(abl)
    avg, mn, gbs = run(0)
    print("ablate %d (0 product, 1 loads only, 2 math only, 4 no x staging) ffn 2x2: %7.2f us (min %6.2f) %7.1f GB/s" % (abl, avg, mn, gbs), flush=True)
L.q4_set_ablate(0)
for kid, nm in ((6, "attention"), (7, "rmsnorm"), (8, "argmax"), (9, "embedding")):
    print("%-10s graph %.2f us per launch" % (nm, tr.bench_kernel_graph(kid, 32, 20)))
avg, mn, gbs = run(5, 32)
print("classifier: %.2f us %.1f GB/s" % (avg, gbs))
tr.close()
