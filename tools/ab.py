#!/usr/bin/env python3
"""A/B of library builds inside ONE gpurun call: boxes differ by several per cent (clocks, power), so two numbers from two
calls say nothing. tools/ab.py [model] [ntok] [rounds] lib1.so lib2.so ... runs `rounds` interleaved passes; each pass is a
fresh process per library doing 1 warm + 4 timed greedy generations (best and median tokens/s)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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

args = sys.argv[1:]
model = args.pop(0) if args and not args[0].endswith(".so") else "7b"
ntok = int(args.pop(0)) if args and args[0].isdigit() else 256
rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
libs = [os.path.abspath(a) for a in args]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.check_output([sys.executable, "-c", ONE, model, str(ntok)], env=dict(os.environ, Q4_LIB_OVERRIDE=l))
        res[l].append(json.loads(out.decode().strip().splitlines()[-1]))
for l in libs:
    print("%-60s %s -n %d  best %.1f  median of medians %.1f  (%s)" % (
        os.path.relpath(l, ROOT), model, ntok, max(x["best"] for x in res[l]), sorted(x["median"] for x in res[l])[len(res[l]) // 2],
        " ".join("%.1f" % x["median"] for x in res[l])), flush=True)
