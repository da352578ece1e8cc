#!/usr/bin/env python3
"""7B -n 2048 tokens/s for split-context attention settings: positions per block and the smallest bin that splits."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()

path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path):
    synth.write_model(path, "7b")
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
tr = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
for chunk, minbin in ((0, 512), (256, 512), (128, 512), (0, 256), (0, 1024), (256, 1024), (0, 512)):
    L.q4_set_attention_split(chunk, minbin)
    tr.generate_ids(prompt, 2048)
    tps = max(tr.generate_ids(prompt, 2048)[1] for _ in range(2))
    t512 = min(tr.generate_ids(prompt, 512)[3] for _ in range(2))
    t256 = min(tr.generate_ids(prompt, 256)[3] for _ in range(2))
    print("chunk %3d, split from bin %4d : -n 2048 %.1f tokens/s (%.4f ms/token); bin 512 alone %.4f ms/token; -n 256 %.1f tok/s" % (
        chunk, minbin, tps, 1e3 / tps, 1e3 * (t512 - t256) / 256, 255 / t256), flush=True)
tr.close()
