#!/usr/bin/env python3
"""Token rate of the -n 256 generation over the tuning knobs of the fused attention-block launch (profiling build)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]


def rate():
    tr.generate_ids(prompt, 256)
    return max(tr.generate_ids(prompt, 256)[1] for _ in range(3))


L.q4_set_fusion(1)
print("fusion 1 (five launches per layer): %.1f tok/s" % rate(), flush=True)
L.q4_set_fusion(2)
for gate in (0,):
    for early in (8, 0, 4, 8 | (4 << 8), 8 | (8 << 8)):
        L.q4_set_gemv_early(4, early)
        print("fusion 2 gate %d early %d hold %d: %.1f tok/s" % (gate, early & 255, early >> 8, rate()), flush=True)
api.check(L.q4_handoff_status(tr.state))
tr.close()
