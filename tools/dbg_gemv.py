import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api, synth
import oracle
L = api.lib(); api.check(L.q4_set_device(0))
def run(K, N, w, z, s, x):
    dw = api.DevQWeight(w, z, s); dx = api.DevBuf(x); do = api.DevBuf(nbytes=N*2)
    api.matmul_q4(do, dx, dw, K, N); api.synchronize()
    return do.get(np.float16, N)
K, N = 256, 8
a, b, c = synth.qweight_sizes(K, N)
for name, qv, zv, xfun in [("q=1,z=0,x=1", 0x11111111, 0, lambda k: np.ones(k)), ("q=1,z=3,x=1", 0x11111111, 0x33333333, lambda k: np.ones(k)),
                           ("q=nib idx,z=0,x=1", 0x76543210, 0, lambda k: np.ones(k)), ("q=1,z=0,x=k%8", 0x11111111, 0, lambda k: (np.arange(k) % 8).astype(float)),
                           ("q=nib idx,z=0,x=(k%8==2)", 0x76543210, 0, lambda k: (np.arange(k) % 8 == 2).astype(float))]:
    w = np.full(a, qv, dtype=np.uint32); z = np.full(b, zv, dtype=np.uint32); s = np.ones(c, dtype=np.float16)
    x = xfun(K).astype(np.float16)
    print(name, "gpu", run(K, N, w, z, s, x)[:4], "ref", oracle.matmul_q4(x, w, z, s, K, N)[:4])
rng = np.random.default_rng(0)
w, z, s = synth.random_qweight(rng, K, N); x = rng.standard_normal(K).astype(np.float16)
print("random gpu", run(K, N, w, z, s, x), "\n       ref", oracle.matmul_q4(x, w, z, s, K, N))
