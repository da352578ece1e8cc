import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
from llama_cu_awq_amd import api, synth
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path): synth.write_model(path, "7b")
L = api.lib(); api.check(L.q4_set_device(0)); s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
t0 = time.time()
for temp, topp, n in ((0.0, 0.9, 2048), (0.5, 0.6, 2048), (1.0, 0.9, 1024), (0.8, 1.0, 512)):
    tr = api.Transformer(path, temperature=temp, topp=topp, seed=42)
    a = tr.generate_ids(prompt, n)
    b = None
    tr.close()
    tr = api.Transformer(path, temperature=temp, topp=topp, seed=42)
    b = tr.generate_ids(prompt, n)
    same = list(a[0]) == list(b[0])
    print("t=%.1f p=%.1f n=%d: %.1f tok/s, %d tokens, reproducible=%s" % (temp, topp, n, b[1], b[2], same), flush=True)
    tr.close()
print("soak seconds", round(time.time() - t0, 1))
