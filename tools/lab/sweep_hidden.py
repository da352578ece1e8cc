#!/usr/bin/env python3
"""Does a launch pay for an uneven number of blocks per CU? Bodies with dim / hidden_dim chosen so that a GEMV's grid becomes a
whole or a fractional number of blocks per CU (256 CUs): per-launch time inside a hipGraph over a ring of 8 layers, bytes / time.
tools/lab/sweep_hidden.py gateup|down13b"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "gateup"
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
# gate/up of the 7B body: 8 columns per block -> hidden / 8 / 256 blocks per CU; down of the 13B body: dim / 8 / 256
cases = [(4096, h) for h in (10240, 10752, 11008, 11264, 12288)] if what == "gateup" else [(d, 13824) for d in (4096, 4608, 5120, 5632, 6144)]
for dim, hidden in cases:
    path = "/tmp/llama2_q4_synth_d%d_h%d.bin" % (dim, hidden)
    geom = (dim, hidden, 8, dim // 128, dim // 128, 512, 64, 10000.0)
    if not os.path.exists(path):
        synth.write_model(path, geom)
    tr = api.Transformer(path)
    tr.reset([1, 5, 9])
    tr.run_transformer(False)
    api.synchronize()
    res = []
    for kid, name, nbytes, blocks in ((0, "gate/up", 2 * sum(synth.qweight_sizes(dim, hidden)[i] * (4, 4, 2)[i] for i in range(3)), hidden / 8),
                                      (2, "down", sum(synth.qweight_sizes(hidden, dim)[i] * (4, 4, 2)[i] for i in range(3)), dim / 8)):
        us = min(tr.bench_kernel_graph(kid, 32, 20) for _ in range(3))
        res.append("%s %.3f blocks per CU %.2f us %.0f GB/s" % (name, blocks / 256.0, us, nbytes / us / 1e3))
    print("dim %5d hidden %5d: %s" % (dim, hidden, ";  ".join(res)), flush=True)
    tr.close()
    os.remove(path)
