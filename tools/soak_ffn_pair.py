#!/usr/bin/env python3
"""Soak of fusion level 4 (the FFN half of a layer as one launch, csrc/gemv_ffn_pair.h) on the 7B geometry: N greedy generations must reproduce the token
ring of fusion level 3 exactly (any stale, torn or missed granule of the in-launch hand-off would show as a difference) with no bounded wait running out.
tools/soak_ffn_pair.py [generations of -n 256] [generations of -n 2048] [level: 5 (with the next layer's QKV as third phase) or 4]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth
n256 = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n2048 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
LEVEL = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path): synth.write_model(path, "7b")
t = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
t0 = time.time()
for n, reps in ((256, n256), (2048, n2048)):
    L.q4_set_fusion(3)
    want = list(t.generate_ids(prompt, n)[0])
    r3 = [t.generate_ids(prompt, n)[1] for _ in range(3)]
    L.q4_set_fusion(LEVEL)
    rates = []
    for r in range(reps):
        toks, tps, timed, secs = t.generate_ids(prompt, n)
        rates.append(tps)
        assert list(toks) == want, (n, r, int(np.argmax(np.array(toks) != np.array(want))))
        assert L.q4_get_fusion() == LEVEL and L.q4_handoff_timeouts() == 0, (n, r)
    api.check(L.q4_handoff_status(t.state))
    print("-n %d: %d generations at level %d equal level 3's ring; tokens/s level 3 best %.1f | min %.1f median %.1f max %.1f; time-outs %d" % (
        n, reps, LEVEL, max(r3), min(rates[1:]), float(np.median(rates[1:])), max(rates[1:]), L.q4_handoff_timeouts()), flush=True)
print("kv stream price: %.4f ticks of 10 ns per position (nominal %.4f)" % (L.q4_kv_stream_price(t.state), 16384 / 5.9e12 * 1e8))
print("soak %.0f s PASS" % (time.time() - t0))
t.close()
