#!/usr/bin/env python3
"""Does an int4 GEMV of the HIP path carry a SYSTEMATIC (signed) error? tools/error_growth.py shows the residual stream's mean error drifting
by about -4.6e-5 per layer, entering in the FFN half. Mean signed error (output - exact) over many trials and outputs, HIP kernel and CPU
restatement, for the accumulating down projection / o-proj and their inputs' distributions.   tools/bias_probe.py [trials]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api as q4, synth   # noqa: E402
import oracle as orc                             # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = q4.lib()
q4.check(L.q4_set_device(0))
s = C.c_void_p()
q4.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
rng = np.random.default_rng(11)


def h(a):
    return np.asarray(a, np.float64).astype(np.float16)


def silu_like(n):          # the distribution of hb = silu(g) * u for g, u ~ N(0, 0.9)
    g, u = rng.standard_normal(n) * 0.9, rng.standard_normal(n) * 0.9
    return h(g / (1 + np.exp(-g)) * u)


cases = [
    ("down 11008->4096 accum, x = silu(g)*u", 11008, 4096, True, lambda: silu_like(11008), 9.0),
    ("down 11008->4096 accum, x ~ N(0, 0.3)", 11008, 4096, True, lambda: h(rng.standard_normal(11008) * 0.3), 9.0),
    ("down 11008->4096 plain, x = silu(g)*u", 11008, 4096, False, lambda: silu_like(11008), 0.0),
    ("o-proj 4096->4096 accum, x ~ N(0, 0.15)", 4096, 4096, True, lambda: h(rng.standard_normal(4096) * 0.15), 9.0),
    ("gemv 4096->11008 plain, x ~ N(0, 1)", 4096, 11008, False, lambda: h(rng.standard_normal(4096)), 0.0),
    ("gemv 4096->4096 plain, x ~ N(0.05, 1) (a mean)", 4096, 4096, False, lambda: h(rng.standard_normal(4096) + 0.05), 0.0),
]
for name, K, N, accum, mkx, oldscale in cases:
    eg, er, mags = [], [], []
    for t in range(T):
        w, z, sc = synth.random_qweight(rng, K, N)
        x = mkx()
        old = h(rng.standard_normal(N) * oldscale) if accum else None
        y64 = orc.matmul_q4_f64(x, w, z, sc, K, N)
        ex = y64 + (old.astype(np.float64) if accum else 0.0)
        rest = orc.matmul_q4(x, w, z, sc, K, N, accum_into=old)
        dw = q4.DevQWeight(w, z, sc)
        dx, dout = q4.DevBuf(x), q4.DevBuf(old if accum else np.zeros(N, np.float16))
        q4.matmul_q4(dout, dx, dw, K, N, accum=accum)
        q4.synchronize()
        g = dout.get(np.float16, N).astype(np.float64)
        eg.append(g - ex)
        er.append(rest.astype(np.float64) - ex)
        mags.append(np.sqrt(np.mean(y64 ** 2)))
    eg, er = np.concatenate(eg), np.concatenate(er)
    n = len(eg)
    print("%-50s sum rms %.3f | mean signed error HIP %+.3e +- %.1e   restatement %+.3e +- %.1e | rms HIP %.3e rest %.3e" % (
        name, float(np.mean(mags)), eg.mean(), eg.std() / np.sqrt(n), er.mean(), er.std() / np.sqrt(n), np.sqrt(np.mean(eg ** 2)), np.sqrt(np.mean(er ** 2))), flush=True)
