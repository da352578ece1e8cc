#!/usr/bin/env python3
"""Rounding quality of the int4 GEMV: rms error (in fp16 ulps of the result) of the HIP kernel and of the reference-order
CPU restatement against an fp64 evaluation of the same sum."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api as q4, synth   # noqa: E402
import oracle as orc                             # noqa: E402

q4.check(q4.lib().q4_set_device(0))
import ctypes as C
s = C.c_void_p(); q4.check(q4.lib().q4_stream_create(C.byref(s))); q4.lib().q4_set_stream(s)
rng = np.random.default_rng(1)
for K, N in ((4096, 11008), (11008, 4096), (5120, 13824)):
    w, z, sc = synth.random_qweight(rng, K, N)
    x = rng.standard_normal(K).astype(np.float16)
    ref16 = orc.matmul_q4(x, w, z, sc, K, N).astype(np.float64)
    ref64 = orc.matmul_q4_f64(x, w, z, sc, K, N)
    dw = q4.DevQWeight(w, z, sc); dx, dout = q4.DevBuf(x), q4.DevBuf(nbytes=N * 2)
    q4.matmul_q4(dout, dx, dw, K, N); q4.synchronize()
    g = dout.get(np.float16, N).astype(np.float64)
    ulp = np.spacing(np.abs(ref64).astype(np.float16)).astype(np.float64)
    eg, eo = (g - ref64) / ulp, (ref16 - ref64) / ulp
    big = np.abs(ref64) > 0.25 * np.abs(ref64).std()
    cr = ref64.astype(np.float16).astype(np.float64)           # correctly rounded result
    print("        |y| > 0.25 sigma (%d of %d): rms ulps HIP %.4f restatement %.4f; abs rms err HIP %.3e restatement %.3e (sigma_y %.3f); "
          "not correctly rounded: HIP %.4f restatement %.4f" % (big.sum(), N, np.sqrt(np.mean(eg[big] ** 2)), np.sqrt(np.mean(eo[big] ** 2)),
           np.sqrt(np.mean((g - ref64) ** 2)), np.sqrt(np.mean((ref16 - ref64) ** 2)), ref64.std(), np.mean(g != cr), np.mean(ref16 != cr)))
    print("%5d x %5d: rms error in fp16 ulps  HIP %.4f  restatement %.4f   (ideal rounding 0.2887)   max %.2f / %.2f" %
          (K, N, np.sqrt(np.mean(eg ** 2)), np.sqrt(np.mean(eo ** 2)), np.abs(eg).max(), np.abs(eo).max()))
