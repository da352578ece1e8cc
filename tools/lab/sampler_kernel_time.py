#!/usr/bin/env python3
"""Microseconds per q4_sample call (topp_sample_kernel alone, vocabulary 32000) for the CLI's sampler settings, on logits shaped
like the synthetic 7B model's (N(0, 1.28)) and on a peaked, trained-model-like row.  tools/lab/sampler_kernel_time.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

PROF = os.environ.get("Q4_PROFILING_BUILD", "") == "1"     # then the kernel stamps its phases into the sampler's indices scratch

path = "/tmp/llama2_q4_synth_v32k_seed5.bin"
if not os.path.exists(path):
    synth.write_model(path, "v32k", seed=5)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
rng = np.random.default_rng(3)
rows = {"N(0,1.28)": (rng.standard_normal(32000) * 1.28).astype(np.float16)}
peaked = (rng.standard_normal(32000) * 2.0).astype(np.float16)
peaked[rng.integers(0, 32000, 20)] += np.float16(9.0)
rows["peaked"] = peaked
for name, logits in rows.items():
    for temp, topp in ((0.5, 0.6), (1.0, 0.9), (1.0, 1.0), (0.02, 0.9)):
        t = api.Transformer(path, temperature=temp, topp=topp, seed=11)
        t.reset([1])
        n = 200
        us = {}
        for what in ("copy", "copy+sample"):
            for rep in range(2):
                api.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    if (i & 31) == 0:
                        t.reset([1])
                    # the sampler normalises IN PLACE (like the reference's softmax_logits_kernel): fresh logits every call
                    api.check(L.q4_memcpy_h2d(t.state.contents.logits, logits.ctypes.data, logits.nbytes))
                    if what != "copy":
                        api.check(L.q4_sample(t.sampler, t.state, 1))
                api.synchronize()
                us[what] = (time.perf_counter() - t0) / n * 1e6
        us = us["copy+sample"] - us["copy"]
        print("%-10s temperature %.2f top-p %.1f: %.1f us per sample() call (last token %d)" % (name, temp, topp, us, t.token(t.pos())), flush=True)
        if PROF:
            class SamplerStruct(C.Structure):
                _fields_ = [("vocab_size", C.c_int), ("indices", C.c_void_p)]
            ind = C.cast(t.sampler, C.POINTER(SamplerStruct)).contents.indices
            st = np.empty(8, dtype=np.uint64)
            api.check(L.q4_memcpy_d2h(st.ctypes.data, ind, st.nbytes))
            d = (st[1:7].astype(np.int64) - st[0:6].astype(np.int64)) * 0.01
            print("           phases (us): load + divide + local max %.2f | block max %.2f | exp + local sum %.2f | block sum %.2f | "
                  "normalise + store + top %.2f | search / sort / scan %.2f" % tuple(d), flush=True)
        t.close()
