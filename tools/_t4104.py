import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from llama_cu_awq_amd import api, synth
api.use_profiling_build()
L = api.lib(); api.check(L.q4_set_device(0))
rng = np.random.default_rng(3)
K=4096
for N in (4096, 6144, 7168, 8192):
    x = rng.standard_normal(K).astype(np.float16)
    g, u = synth.random_qweight(rng, K, N), synth.random_qweight(rng, K, N)
    dg, du, dx = api.DevQWeight(*g), api.DevQWeight(*u), api.DevBuf(x)
    ref=None
    for e in (0, 8, 9, 10, 12, 13, 14):
        L.q4_set_gemv_early(11, e)
        bad = 0
        for rep in range(30):
            do = api.DevBuf(nbytes=N*2); api.ffn_matvec_silu(do, dx, dg, du, K, N); api.synchronize()
            o = do.get(np.uint16, N)
            if ref is None: ref = o
            elif not (o==ref).all(): bad += 1
        print("N", N, "engine", e, "bad reps", bad, "of 30", flush=True)
