#!/usr/bin/env python3
"""Where does the HIP path's distance from exact arithmetic enter? For positions 0..N-1 of the 7B geometry the residual stream x is
dumped after the attention half (o-proj + residual) and after the FFN half (down projection + residual) of every layer -- on the GPU
(profiling build, eager launches, q4_set_layer_dump), in the CPU restatement of the reference's arithmetic (oracle orc_forward) and in
the never-rounded double forward (orc_forward_f64) -- at fusion levels 3, 1 and 0, all fed the SAME tokens. Printed per half layer:
rms and max over the positions of (GPU x - f64 x) and (restatement x - f64 x), and their ratio.
tools/error_growth.py [model] [positions]  ->  gpurun_out/error_growth_<model>.json (+ the table on stdout)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llama_cu_awq_amd import api, synth   # noqa: E402
import oracle                              # noqa: E402  (a measurement tool: the checker, never the product)

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
npos = int(sys.argv[2]) if len(sys.argv) > 2 else 32
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
PROMPT = [1, 2436, 385, 3686, 388, 1048, 22796, 118]

# ---- CPU: restatement + exact, same tokens (the restatement's greedy continuation) ----------------------------------
m = oracle.Model(path)
d16, d64 = m.layer_dump()
nl, dim = m.cfg.n_layers, m.cfg.dim
toks = list(PROMPT)
rest = np.zeros((npos, nl, 2, dim), np.float64)
exact = np.zeros((npos, nl, 2, dim), np.float64)
rest_logits, exact_logits = [], []
for pos in range(npos):
    lg = m.forward(toks[pos], pos)
    rest[pos] = d16.astype(np.float64)
    rest_logits.append(lg.astype(np.float64))
    if pos + 1 >= len(toks):
        toks.append(int(np.argmax(lg.astype(np.float32))))
    exact_logits.append(m.forward_f64(toks[pos], pos, cap=npos))
    exact[pos] = d64
    print("cpu position %d done" % pos, file=sys.stderr, flush=True)
# control: the SAME arithmetic in another valid fp32 order (oracle orc_set_order_variant): how two legitimate evaluations of the reference compare
oracle.lib().orc_set_order_variant(1)
ctrl = np.zeros((npos, nl, 2, dim), np.float64)
for pos in range(npos):
    m.forward(toks[pos], pos)
    ctrl[pos] = d16.astype(np.float64)
oracle.lib().orc_set_order_variant(0)
m.close()


def describe(name, e_a, e_b):
    """e_a, e_b: errors of two evaluations against the exact forward, [npos, nl, 2, dim]"""
    print("%s: per half layer -- rms(a) / rms(b), correlation of the two error vectors, rms(a - b) / rms(b), mean signed error a | b" % name)
    for l in (0, 1, 3, 7, 15, 23, nl - 1):
        for hh in (0, 1):
            a, b = e_a[:, l, hh].ravel(), e_b[:, l, hh].ravel()
            ra, rb = np.sqrt(np.mean(a * a)), np.sqrt(np.mean(b * b))
            print("   layer %2d %-9s ratio %.4f  corr %.4f  |a-b|/|b| %.4f  mean %+.3e | %+.3e" % (
                l, "attention" if hh == 0 else "ffn", ra / rb, float(np.mean(a * b) / (ra * rb)), float(np.sqrt(np.mean((a - b) ** 2)) / rb), a.mean(), b.mean()))


describe("restatement in another valid fp32 order (a) vs restatement (b)", ctrl - exact, rest - exact)

# ---- GPU: every fusion level, eager, the same token list fed as a prompt ---------------------------------------------
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
L.q4_set_use_graphs(0)
L.q4_set_layer_dump.argtypes = [C.c_void_p]
out = {"model": model, "positions": npos, "levels": {}}
for level in (3, 1, 0):
    L.q4_set_fusion(level)
    tr = api.Transformer(path)
    dump = api.DevBuf(nbytes=nl * 2 * dim * 2)
    L.q4_set_layer_dump(dump.ptr)
    tr.reset(toks[:npos + 1])
    gpu = np.zeros((npos, nl, 2, dim), np.float64)
    gl = []
    for pos in range(npos):
        tr.run_transformer(False)
        api.synchronize()
        gpu[pos] = dump.get(np.float16).reshape(nl, 2, dim).astype(np.float64)
        gl.append(tr.logits().astype(np.float64))
    L.q4_set_layer_dump(None)
    tr.close()
    eg, er = gpu - exact, rest - exact
    describe("fusion level %d: GPU (a) vs restatement (b)" % level, eg, er)
    describe("fusion level %d: GPU (a) vs restatement in the other order (b)" % level, eg, ctrl - exact)
    rows = []
    for l in range(nl):
        for h in range(2):
            g_rms = float(np.sqrt(np.mean(eg[:, l, h] ** 2)))
            r_rms = float(np.sqrt(np.mean(er[:, l, h] ** 2)))
            rows.append({"layer": l, "half": "attention" if h == 0 else "ffn", "gpu_rms": g_rms, "rest_rms": r_rms,
                         "gpu_max": float(np.abs(eg[:, l, h]).max()), "rest_max": float(np.abs(er[:, l, h]).max()),
                         "ratio_rms": g_rms / r_rms if r_rms else None,
                         "x_rms": float(np.sqrt(np.mean(exact[:, l, h] ** 2)))})
    gl, rl, el = np.array(gl), np.array(rest_logits), np.array(exact_logits)
    den = np.maximum(1.0, np.abs(el))
    lg = {"gpu_rms": float(np.sqrt(np.mean((gl - el) ** 2))), "rest_rms": float(np.sqrt(np.mean((rl - el) ** 2))),
          "gpu_max_rel": float(np.max(np.abs(gl - el) / den)), "rest_max_rel": float(np.max(np.abs(rl - el) / den)),
          "gpu_vs_rest_max_rel": float(np.max(np.abs(gl - rl) / np.maximum(1.0, np.abs(rl))))}
    # per position: does the ratio depend on the context length?
    per_pos = [{"pos": p, "gpu_rms": float(np.sqrt(np.mean(eg[p, nl - 1, 1] ** 2))), "rest_rms": float(np.sqrt(np.mean(er[p, nl - 1, 1] ** 2)))}
               for p in range(npos)]
    out["levels"][str(level)] = {"half_layers": rows, "logits": lg, "last_layer_by_position": per_pos}
    print("fusion level %d: logits rms GPU %.5f restatement %.5f (ratio %.3f); max rel GPU %.4f restatement %.4f" % (
        level, lg["gpu_rms"], lg["rest_rms"], lg["gpu_rms"] / lg["rest_rms"], lg["gpu_max_rel"], lg["rest_max_rel"]))
    print("  layer half       x rms     GPU rms   rest rms   ratio |  GPU max   rest max")
    for r in rows:
        if r["layer"] in (0, 1, 2, 3, 7, 15, 23, nl - 2, nl - 1):
            print("  %5d %-9s %8.4f  %9.6f  %9.6f  %5.3f | %8.5f  %8.5f" % (r["layer"], r["half"], r["x_rms"], r["gpu_rms"], r["rest_rms"],
                                                                        r["ratio_rms"] or 0, r["gpu_max"], r["rest_max"]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "error_growth_%s.json" % model), "w"), indent=1)
