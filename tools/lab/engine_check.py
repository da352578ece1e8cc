#!/usr/bin/env python3
"""Step A of the loader / consumer engine (csrc/exp/ffn_engine.h) against the shipped gate/up kernel, in ONE process:
bit equality of q4_ffn_matvec_silu and of the norm-fused launch on the 7B geometry, then per-launch time (HIP events
and inside a hipGraph) and tokens/s for knob 11 = -1 (gemv_q4_kernel), 0 (the product's choice), 1..3 (engine, vmcnt lag), 5, 6 (lag 1, 2 with the
consumers' next-slot prefetch), 8..14 (strips variants, csrc/gemv_strip.h).  ENGINE_VARIANTS=1,5 tools/lab/engine_check.py [model]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
ENGINE = 11
VARIANTS = [int(v) for v in os.environ.get("ENGINE_VARIANTS", "0,1,8").split(",")]

# ---- 1. the public op, no norm -------------------------------------------------------------------------
rng = np.random.default_rng(7)
K, N = 4096, 11008
bad = 0
for trial in range(6):
    g = synth.random_qweight(rng, K, N)
    u = synth.random_qweight(rng, K, N)
    x = rng.standard_normal(K).astype(np.float16)
    dg, du = api.DevQWeight(*g), api.DevQWeight(*u)
    dx = api.DevBuf(x)
    outs = {}
    for v in [-1] + VARIANTS:
        L.q4_set_gemv_early(ENGINE, v)
        for rep in range(5):
            do = api.DevBuf(nbytes=N * 2)
            api.ffn_matvec_silu(do, dx, dg, du, K, N)
            api.synchronize()
            o = do.get(np.uint16, N)
            if v == -1 and rep == 0:
                outs[0] = o
            elif not (o == outs[0]).all():
                bad += 1
                d = np.nonzero(o != outs[0])[0]
                print("MISMATCH trial %d engine %d rep %d: %d outputs differ, first at %d (%04x vs %04x)" % (
                    trial, v, rep, d.size, d[0], o[d[0]], outs[0][d[0]]), flush=True)
print("q4_ffn_matvec_silu 4096 -> 11008, 6 weight sets x 5 repeats x engines %s: %s" % (VARIANTS, "bit-identical" if bad == 0 else "%d MISMATCHES" % bad), flush=True)

# ---- 2. inside the model: norm-fused launch, timing ------------------------------------------------------
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
tr.generate_ids(prompt, 16)     # a real residual stream in s->x
st = tr.state.contents
hidden = tr.config.hidden_dim


def hb():
    out = np.empty(hidden, dtype=np.uint16)
    api.check(L.q4_memcpy_d2h(out.ctypes.data, st.hb, out.nbytes))
    return out


ref = None
for v in [-1] + VARIANTS:
    L.q4_set_gemv_early(ENGINE, v)
    for rep in range(4):
        tr.bench_kernel(0, 1)
        o = hb()
        if ref is None:
            ref = o
        elif not (o == ref).all():
            d = np.nonzero(o != ref)[0]
            print("MISMATCH fused launch, engine %d rep %d: %d of %d differ, first %d" % (v, rep, d.size, hidden, d[0]), flush=True)
            bad += 1
print("norm-fused gate/up launch on the model's layer 0: %s" % ("bit-identical" if bad == 0 else "MISMATCH"), flush=True)

for rnd in range(3):
    for v in [-1] + VARIANTS:
        L.q4_set_gemv_early(ENGINE, v)
        tr.bench_kernel(0, 32)
        avg, mn, mx = tr.bench_kernel(0, 256)
        g = min(tr.bench_kernel_graph(0, 32, 20) for _ in range(3))
        print("round %d engine %d: HIP events avg %.2f min %.2f max %.2f us; in a graph %.2f us per launch" % (rnd, v, avg, mn, mx, g), flush=True)

ring = None
for rnd in range(3):
    for v in [-1] + VARIANTS:
        L.q4_set_gemv_early(ENGINE, v)
        tr.generate_ids(prompt, 256)
        r = sorted(tr.generate_ids(prompt, 256)[1] for _ in range(4))
        toks = tr.generate_ids(prompt, 256)[0]
        if ring is None:
            ring = toks.copy()
        print("round %d engine %d: -n 256 best %.1f median %.1f tok/s; tokens equal to engine -1: %s" % (
            rnd, v, r[-1], 0.5 * (r[1] + r[2]), bool((toks == ring).all())), flush=True)
print("handoff timeouts", L.q4_handoff_timeouts(), flush=True)
tr.close()
