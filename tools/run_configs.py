#!/usr/bin/env python3
"""BASELINE.json configs 3 and 4 on synthetic weights: 13B -n 256 and 7B -n 2048 (KV-bandwidth stress), each with a
GPU-vs-CPU-restatement logit check on the first positions. Prints one JSON line per config."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402
import oracle                              # noqa: E402

PROMPT = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
which = sys.argv[1:] or ["13b", "7b-2048"]
for cfg in which:
    model, ntok = ("13b", 256) if cfg == "13b" else ("7b", 2048)
    path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
    if not os.path.exists(path):
        synth.write_model(path, model)
    tr = api.Transformer(path)
    tr.generate_ids(PROMPT, ntok)                       # warm + graph capture of every bin
    res = [tr.generate_ids(PROMPT, ntok) for _ in range(2)]
    tps = max(r[1] for r in res)
    out = {"config": cfg, "tokens_per_s": round(tps, 1), "timed_tokens": res[0][2], "ms_per_token": round(1000.0 / tps, 4)}
    if cfg == "13b":
        m = oracle.Model(path)
        tr.reset(PROMPT)
        worst = 0.0
        for pos in range(6):
            tr.run_transformer(False)
            api.synchronize()
            ref = m.forward(PROMPT[pos], pos).astype(np.float32)
            got = tr.logits().astype(np.float32)
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))))
        out["logits_max_rel_err_vs_cpu_restatement_6pos"] = round(worst, 5)
        m.close()
    print(json.dumps(out), flush=True)
    tr.close()
