#!/usr/bin/env python3
"""Repeatability of the models whose launches run as strips (13B: gate/up pair units, down projection, classifier; Mistral geometry: gate/up, classifier):
N greedy generations must reproduce the first token ring (a race between an LDS-DMA fill and a read would show as a run-to-run difference)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
for model, n, reps in (("13b", 256, 30), ("mistral7b", 256, 20)):
    path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
    if not os.path.exists(path): synth.write_model(path, model)
    t = api.Transformer(path)
    prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
    first = None; rates = []
    for r in range(reps):
        toks, tps, timed, secs = t.generate_ids(prompt, n)
        rates.append(tps)
        if first is None: first = list(toks)
        assert list(toks) == first, (model, r)
    api.check(L.q4_handoff_status(t.state))
    print(model, "rings equal over", reps, "generations; tokens/s min %.1f max %.1f" % (min(rates[1:]), max(rates[1:])), "timeouts", L.q4_handoff_timeouts(), flush=True)
    t.close()
