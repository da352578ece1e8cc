#!/usr/bin/env python3
"""gate/up launch-shape sweep: waves per block (blocks per CU) x early-bird waves per CU (in-graph us per launch)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()   # the measurement knobs live in libllama2_q4_prof.so only

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
hidden = tr.config.hidden_dim
for cols in (2, 4):
    for waves in (4, 8):
        blocks = -(-hidden // (cols * waves))
        for e in (0, waves):
            L.q4_set_gemv_tune(3, cols, waves)
            L.q4_set_gemv_early(3, e)
            try:
                g = min(tr.bench_kernel_graph(0, 32, 20) for _ in range(3))
                print("gate/up cols %d waves %2d (%4d blocks, %.2f per CU) early %2d : %.2f us" % (cols, waves, blocks, blocks / 256.0, e, g), flush=True)
            except Exception as ex:
                print("gate/up cols %d waves %2d early %2d : failed (%s)" % (cols, waves, e, ex), flush=True)
tr.close()
