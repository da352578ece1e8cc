#!/usr/bin/env python3
"""Search-phase time of topp_sample_kernel (profiling build's stamps) over many coins, -t 0.5 -p 0.6, vocabulary 32000, N(0, 1.28) logits."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth
api.use_profiling_build()
path = "/tmp/llama2_q4_synth_v32k_seed5.bin"
if not os.path.exists(path): synth.write_model(path, "v32k", seed=5)
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
rng = np.random.default_rng(3)
logits = (rng.standard_normal(32000) * 1.28).astype(np.float16)
t = api.Transformer(path, temperature=0.5, topp=0.6, seed=11)
class SamplerStruct(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("indices", C.c_void_p)]
ind = C.cast(t.sampler, C.POINTER(SamplerStruct)).contents.indices
res = []
for i in range(64):
    if (i & 15) == 0: t.reset([1])
    api.check(L.q4_memcpy_h2d(t.state.contents.logits, logits.ctypes.data, logits.nbytes))
    api.check(L.q4_sample(t.sampler, t.state, 1)); api.synchronize()
    st = np.empty(8, dtype=np.uint64); api.check(L.q4_memcpy_d2h(st.ctypes.data, ind, st.nbytes))
    res.append(((int(st[5]) - int(st[0])) * 0.01, (int(st[6]) - int(st[5])) * 0.01))
r = np.array(res)
print("normalise: median %.2f us; search: min %.2f p25 %.2f median %.2f p75 %.2f max %.2f us" % ((np.median(r[:, 0]),) + tuple(np.percentile(r[:, 1], [0, 25, 50, 75, 100]))))
print("search times:", " ".join("%.1f" % v for v in r[:, 1]))
t.close()
