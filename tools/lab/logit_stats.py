#!/usr/bin/env python3
"""What the sampler sees: per decode step of the synthetic 7B model at -t 0.5, the count and mass of probabilities >= 2^-e (e = 9 .. 14) -- the candidate
sets of csrc/q4_sampling.hip -- next to the thresholds coin * top-p can take."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path): synth.write_model(path, "7b")
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
t = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
t.reset(prompt)
rows = []
for pos in range(72):
    t.run_transformer(pos >= len(prompt) - 1)
    api.synchronize()
    if pos < 8: continue
    l = t.logits().astype(np.float64) / 0.5
    p = np.exp(l - l.max()); p /= p.sum()
    rows.append([l.std() * 0.5, p.max()] + [v for e in range(9, 15) for v in (int((p >= 2.0 ** -e).sum()), float(p[p >= 2.0 ** -e].sum()))])
r = np.array(rows)
print("logit std %.3f..%.3f  p_max %.4f..%.4f" % (r[:, 0].min(), r[:, 0].max(), r[:, 1].min(), r[:, 1].max()))
for k, e in enumerate(range(9, 15)):
    print("p >= 2^-%d: count %4d..%4d (median %4d)  mass %.3f..%.3f (median %.3f)" % (e, r[:, 2 + 2 * k].min(), r[:, 2 + 2 * k].max(), np.median(r[:, 2 + 2 * k]),
                                                                                  r[:, 3 + 2 * k].min(), r[:, 3 + 2 * k].max(), np.median(r[:, 3 + 2 * k])))
t.close()
