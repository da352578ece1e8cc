#!/usr/bin/env python3
"""Token rate of the product library under HIP runtime settings (environment variables read when the runtime initialises), one fresh
process per setting, interleaved rounds, ONE gpurun call.  tools/lab/env_sweep.py [model] [ntok] [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ONE = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, %r)
from llama_cu_awq_amd import api, synth
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
model, n = sys.argv[1], int(sys.argv[2])
path = "/tmp/llama2_q4_synth_%%s_seed20240229.bin" %% model
if not os.path.exists(path): synth.write_model(path, model)
t = api.Transformer(path)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
t.generate_ids(prompt, n)
r = sorted(t.generate_ids(prompt, n)[1] for _ in range(4))
print(json.dumps({"best": r[-1], "median": 0.5 * (r[1] + r[2])}))
t.close()
''' % ROOT

SETTINGS = [
    {},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    {"HIP_FORCE_DEV_KERNARG": "0"},
    {"HIP_FORCE_DEV_KERNARG": "1"},
    {"AMD_OPT_FLUSH": "0"},
    {"AMD_OPT_FLUSH": "1"},
    {"DEBUG_HIP_KERNARG_COPY_OPT": "0"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1024"},
    {"GPU_MAX_HW_QUEUES": "1"},
    {"AMD_DIRECT_DISPATCH": "0"},
    {"DEBUG_CLR_KERNARG_HDP_FLUSH_WA": "0"},
    {"HSA_ENABLE_SDMA": "0"},
    {"GPU_FLUSH_ON_EXECUTION": "1"},
]
args = sys.argv[1:]
model = args.pop(0) if args else "7b"
ntok = int(args.pop(0)) if args else 256
rounds = int(args.pop(0)) if args else 2
res = [[] for _ in SETTINGS]
for r in range(rounds):
    for k, st in enumerate(SETTINGS):
        try:
            out = subprocess.check_output([sys.executable, "-c", ONE, model, str(ntok)], env=dict(os.environ, **st), stderr=subprocess.STDOUT, timeout=300)
            res[k].append(json.loads(out.decode().strip().splitlines()[-1]))
        except Exception as ex:   # a setting the runtime rejects
            res[k].append({"best": 0.0, "median": 0.0})
            print("setting %s failed: %s" % (st, str(ex)[:200]), flush=True)
for st, rr in zip(SETTINGS, res):
    print("%-46s %s -n %d  best %.1f  medians %s" % (" ".join("%s=%s" % kv for kv in st.items()) or "(default)", model, ntok, max(x["best"] for x in rr),
                                                  " ".join("%.1f" % x["median"] for x in rr)), flush=True)
