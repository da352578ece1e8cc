#!/usr/bin/env python3
"""BASELINE.json configs 3, 4 and 5 on synthetic weights: 13B -n 256, 7B -n 2048 (KV-bandwidth stress) and the
perplexity path on a 7B-geometry model (64 synthetic tokens: per-position fp32 logits and the perplexity against the CPU
restatement and the unrounded double forward). Prints one JSON line per config."""
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
which = sys.argv[1:] or ["13b", "7b-2048", "7b-ppl"]
for cfg in which:
    model, ntok = ("13b", 256) if cfg == "13b" else ("mistral7b", 256) if cfg == "mistral7b" else ("7b", 2048)
    path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
    if not os.path.exists(path):
        synth.write_model(path, model)
    if cfg == "7b-ppl":                                   # perplexity.h:57-97 on token ids
        npos = 64
        rng = np.random.default_rng(5)
        toks = np.concatenate([[1], rng.integers(3, 32000, size=npos)]).astype(np.int32)
        tr = api.Transformer(path, perplexity=True)
        tr.perplexity_ids(toks)
        t0 = time.perf_counter()
        ppl = tr.perplexity_ids(toks)
        secs = time.perf_counter() - t0
        glog = tr.logits_array(npos).astype(np.float64)
        m = oracle.Model(path)
        rlog = np.stack([m.forward(int(toks[i]), i).astype(np.float64) for i in range(npos)])
        n64 = 16
        elog = np.stack([m.forward_f64(int(toks[i]), i, cap=n64) for i in range(n64)])
        rppl = oracle.compute_perplexity(toks[1:npos + 1], rlog.astype(np.float32))
        den = np.maximum(1.0, np.abs(rlog))
        d64 = np.maximum(1.0, np.abs(elog))
        print(json.dumps({"config": cfg, "positions": npos, "perplexity_gpu": round(float(ppl), 3), "perplexity_cpu_restatement": round(float(rppl), 3),
                          "perplexity_rel_diff": round(abs(float(ppl) - float(rppl)) / float(rppl), 6),
                          "logits_max_rel_err_vs_cpu_restatement": round(float(np.max(np.abs(glog - rlog) / den)), 5),
                          "first_%d_positions_vs_unrounded_f64" % n64: {"gpu": round(float(np.max(np.abs(glog[:n64] - elog) / d64)), 5),
                                                                        "cpu_restatement": round(float(np.max(np.abs(rlog[:n64] - elog) / d64)), 5)},
                          "gpu_seconds_for_%d_steps" % npos: round(secs, 4)}), flush=True)
        m.close()
        tr.close()
        continue
    tr = api.Transformer(path)
    tr.generate_ids(PROMPT, ntok)                       # warm + graph capture of every bin
    res = [tr.generate_ids(PROMPT, ntok) for _ in range(2)]
    tps = max(r[1] for r in res)
    out = {"config": cfg, "tokens_per_s": round(tps, 1), "timed_tokens": res[0][2], "ms_per_token": round(1000.0 / tps, 4)}
    if cfg in ("13b", "mistral7b"):
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
