import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from llama_cu_awq_amd import api, synth
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
for m in ("7b", "head128"):
    path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % m
    if not os.path.exists(path): synth.write_model(path, m)
    t = api.Transformer(path)
    print(m, "price", L.q4_kv_stream_price(t.state))
    t.close()
