#!/usr/bin/env python3
"""Rounding quality of every operation of a decode layer, one at a time, on IDENTICAL fp16 inputs: the HIP kernel (through the C ABI, the 1:1
entry points of fusion level 0) and the CPU restatement of the reference's arithmetic, each against a double-precision evaluation of the same
operation on the same inputs. Printed per operation: rms and max error in fp16 ulps of the exact result, and the absolute rms error.
Companion of tools/error_growth.py (which shows the HIP path's distance from the exact forward growing ~7 % faster than the restatement's
at every fusion level): this names the operation, or shows that no single one differs.   tools/op_accuracy.py [dim] [hidden] [heads] [pos]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api as q4, synth   # noqa: E402
import oracle as orc                             # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 11008
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
pos = int(sys.argv[4]) if len(sys.argv) > 4 else 100
hs = dim // heads
L = q4.lib()
q4.check(L.q4_set_device(0))
s = C.c_void_p()
q4.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
rng = np.random.default_rng(5)
rows = []


def report(name, gpu, rest, exact):
    exact = np.asarray(exact, np.float64).ravel()
    g = np.asarray(gpu, np.float64).ravel()
    r = np.asarray(rest, np.float64).ravel()
    ulp = np.spacing(np.abs(exact).astype(np.float16)).astype(np.float64)
    eg, er = (g - exact) / ulp, (r - exact) / ulp
    row = {"op": name, "gpu_rms_ulp": float(np.sqrt(np.mean(eg ** 2))), "rest_rms_ulp": float(np.sqrt(np.mean(er ** 2))),
           "gpu_max_ulp": float(np.abs(eg).max()), "rest_max_ulp": float(np.abs(er).max()),
           "gpu_abs_rms": float(np.sqrt(np.mean((g - exact) ** 2))), "rest_abs_rms": float(np.sqrt(np.mean((r - exact) ** 2))),
           "gpu_bias_ulp": float(np.mean(eg)), "rest_bias_ulp": float(np.mean(er)), "differ": float(np.mean(g != r))}
    rows.append(row)
    print("%-34s rms ulp HIP %.4f rest %.4f | abs rms HIP %.3e rest %.3e (ratio %.3f) | bias ulp %+.4f / %+.4f | max %.2f / %.2f | outputs that differ %.4f" % (
        name, row["gpu_rms_ulp"], row["rest_rms_ulp"], row["gpu_abs_rms"], row["rest_abs_rms"], row["gpu_abs_rms"] / row["rest_abs_rms"],
        row["gpu_bias_ulp"], row["rest_bias_ulp"], row["gpu_max_ulp"], row["rest_max_ulp"], row["differ"]), flush=True)


def h(a):
    return np.asarray(a, np.float64).astype(np.float16)


# ---- inputs of the magnitudes a mid-depth layer sees (tools/error_growth.py: x rms ~ 9 at layer 15)
x = h(rng.standard_normal(dim) * 9.0)
w_rms = h(1.0 + 0.1 * rng.standard_normal(dim))

# rmsnorm
ex = x.astype(np.float64) * (w_rms.astype(np.float64) / np.sqrt(np.mean(x.astype(np.float64) ** 2) + 1e-5))
dx, dw, do = q4.DevBuf(x), q4.DevBuf(w_rms), q4.DevBuf(nbytes=dim * 2)
q4.rmsnorm(do, dx, dw, dim)
q4.synchronize()
xb_g = do.get(np.float16, dim)
xb_r = orc.rmsnorm(x, w_rms)
report("rmsnorm", xb_g, xb_r, ex)
xb = xb_r.copy()

# q / k / v GEMVs (plain int4 GEMV, K = dim) and the accumulating o-proj
for name, K, N, accum in (("gemv dim->dim (q/k/v)", dim, dim, False), ("gemv dim->dim accum (o-proj)", dim, dim, True),
                          ("gemv hidden->dim accum (down)", hidden, dim, True), ("gemv dim->hidden", dim, hidden, False)):
    w, z, sc = synth.random_qweight(rng, K, N)
    xin = xb if K == dim else h(rng.standard_normal(K) * 0.15)         # hb ~ silu(g) * u: small
    old = h(rng.standard_normal(N) * 9.0)
    ex = orc.matmul_q4_f64(xin, w, z, sc, K, N) + (old.astype(np.float64) if accum else 0.0)
    rest = orc.matmul_q4(xin, w, z, sc, K, N, accum_into=old if accum else None)
    dwq = q4.DevQWeight(w, z, sc)
    dxi, dout = q4.DevBuf(xin), q4.DevBuf(old if accum else np.zeros(N, np.float16))
    q4.matmul_q4(dout, dxi, dwq, K, N, accum=accum)
    q4.synchronize()
    report(name, dout.get(np.float16, N), rest, ex)

# RoPE
qv, kv = h(rng.standard_normal(dim) * 1.2), h(rng.standard_normal(dim) * 1.2)
theta = 10000.0
i = np.arange(hs // 2)
freq = 1.0 / theta ** (2.0 * i / hs)
c, sn = np.cos(pos * freq), np.sin(pos * freq)


def rope64(v):
    v = v.astype(np.float64).reshape(heads, hs)
    a, b = v[:, : hs // 2], v[:, hs // 2:]
    return np.concatenate([a * c - b * sn, a * sn + b * c], axis=1).ravel()


kcache = np.zeros((pos + 1, dim), np.float16)      # the kernel rotates the key row INSIDE the cache, at row `pos` (gpu_kernels.h:347-354)
kcache[pos] = kv
dq, dk, dpos = q4.DevBuf(qv), q4.DevBuf(kcache), q4.DevBuf(np.array([pos], np.int32))
q4.RoPERotation(dq, dk, heads, heads, hs, dpos, 0, theta)
q4.synchronize()
rq, rk = orc.rope(qv, kv, heads, heads, hs, pos, theta)
report("rope (q)", dq.get(np.float16, dim), rq, rope64(qv))
report("rope (k, in the cache)", dk.get(np.float16).reshape(pos + 1, dim)[pos], rk, rope64(kv))

# attention over pos + 1 positions
kc, vc = h(rng.standard_normal((pos + 1, dim)) * 1.2), h(rng.standard_normal((pos + 1, dim)) * 0.6)
qq = rq
q64 = qq.astype(np.float64).reshape(heads, hs)
k64, v64 = kc.astype(np.float64).reshape(pos + 1, heads, hs), vc.astype(np.float64).reshape(pos + 1, heads, hs)
sc64 = np.einsum("hd,thd->ht", q64, k64) / np.sqrt(hs)
p64 = np.exp(sc64 - sc64.max(axis=1, keepdims=True))
p64 /= p64.sum(axis=1, keepdims=True)
ex = np.einsum("ht,thd->hd", p64, v64).ravel()
seq = 128 if pos < 128 else 256 if pos < 256 else 512 if pos < 512 else 2048
kcp, vcp = np.zeros((seq, dim), np.float16), np.zeros((seq, dim), np.float16)
kcp[: pos + 1], vcp[: pos + 1] = kc, vc
dq2, dkc, dvc, dout = q4.DevBuf(qq), q4.DevBuf(kcp), q4.DevBuf(vcp), q4.DevBuf(nbytes=dim * 2)
datt = q4.DevBuf(nbytes=max(heads * seq, heads * 8 * (hs + 4) * 2) * 4)
q4.MultiHeadAttention(dout, dq2, dkc, dvc, datt, heads, hs, 1, seq, dpos)
q4.synchronize()
rest, _ = orc.attention(qq, kcp, vcp, heads, hs, 1, pos, seq)
report("attention (%d positions)" % (pos + 1), dout.get(np.float16, dim), rest, ex)

# gate / up + SiLU
g, u = synth.random_qweight(rng, dim, hidden), synth.random_qweight(rng, dim, hidden)
g64, u64 = orc.matmul_q4_f64(xb, *g, dim, hidden), orc.matmul_q4_f64(xb, *u, dim, hidden)
ex = g64 / (1.0 + np.exp(-g64)) * u64
dg, du = q4.DevQWeight(*g), q4.DevQWeight(*u)
dxb, dout = q4.DevBuf(xb), q4.DevBuf(nbytes=hidden * 2)
q4.ffn_matvec_silu(dout, dxb, dg, du, dim, hidden)
q4.synchronize()
report("ffn gate/up + SiLU", dout.get(np.float16, hidden), orc.ffn_matvec_silu(xb, g, u, dim, hidden), ex)

# classifier (fp16 GEMV)
vocab = 32000
wc = h(rng.standard_normal((vocab, dim)) * 0.02)
ex = wc.astype(np.float64) @ xb.astype(np.float64)
dwc, dout = q4.DevBuf(wc), q4.DevBuf(nbytes=vocab * 2)
q4.matmul(dout, q4.DevBuf(xb), dwc, dim, vocab)
q4.synchronize()
report("classifier fp16 GEMV", dout.get(np.float16, vocab), orc.matmul_f16(xb, wc.ravel(), dim, vocab), ex)

import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"dim": dim, "hidden": hidden, "heads": heads, "pos": pos, "ops": rows}, open("gpurun_out/op_accuracy.json", "w"), indent=1)
