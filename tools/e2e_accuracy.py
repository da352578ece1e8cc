#!/usr/bin/env python3
"""End-to-end rounding quality: rms logit error of the HIP path and of the reference-order CPU restatement against the
unrounded double forward, position by position, on a synthetic model (default: small)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api as q4, synth   # noqa: E402
import oracle as orc                             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "small"
npos = int(sys.argv[2]) if len(sys.argv) > 2 else 48
path = "/tmp/e2e_%s.bin" % name
synth.write_model(path, name, seed=11)
q4.check(q4.lib().q4_set_device(0))
s = C.c_void_p(); q4.check(q4.lib().q4_stream_create(C.byref(s))); q4.lib().q4_set_stream(s)
fus = int(sys.argv[3]) if len(sys.argv) > 3 else 1
q4.lib().q4_set_fusion(fus)
t = q4.Transformer(path)
m = orc.Model(path)
rng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 3)
toks = [1] + [int(v) for v in rng.integers(3, t.config.vocab_size, size=npos - 1)]
t.reset(toks)
rg, rr = [], []
for pos, tok in enumerate(toks):
    t.run_transformer(False); q4.synchronize()
    g = t.logits().astype(np.float64)
    r = m.forward(tok, pos).astype(np.float64)
    e = m.forward_f64(tok, pos, cap=npos)
    rg.append(np.sqrt(np.mean((g - e) ** 2))); rr.append(np.sqrt(np.mean((r - e) ** 2)))
rg, rr = np.array(rg), np.array(rr)
print("%s, fusion %d, %d positions: rms logit error vs unrounded f64: HIP mean %.5f, restatement mean %.5f, ratio of means %.3f, "
      "HIP larger at %d positions" % (name, fus, npos, rg.mean(), rr.mean(), rg.mean() / rr.mean(), int((rg > rr).sum())))
