#!/usr/bin/env python3
"""Soak of the default path (in-launch hand-offs): repeated -n 2048 / -n 256 generations on the 7B and 13B geometries; every run
must reproduce the first run's token ring and leave the hand-off error flag clear. tools/soak.py [seconds]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
t_end = time.time() + budget
for model, n in (("7b", 2048), ("13b", 256), ("7b", 256)):
    path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
    if not os.path.exists(path):
        synth.write_model(path, model)
    tr = api.Transformer(path)
    ref = tr.generate_ids(prompt, n)[0].copy()
    runs, slowest, fastest = 0, 0.0, 1e9
    stop = time.time() + budget / 3
    while time.time() < stop:
        toks, tps, timed, secs = tr.generate_ids(prompt, n)
        assert np.array_equal(toks, ref), "token ring changed in run %d" % runs
        api.check(L.q4_handoff_status(tr.state))
        runs += 1
        slowest, fastest = max(slowest, secs), min(fastest, secs)
    print("%s -n %d: %d runs identical, hand-off flag clear, %.1f .. %.1f tok/s" % (model, n, runs, timed / slowest, timed / fastest), flush=True)
    tr.close()
