#!/usr/bin/env python3
"""tools/op_accuracy.py on the REAL activations of the 7B-geometry model: at position P the restatement's own forward supplies the inputs of
every operation of layer l (its residual stream, its KV cache); each operation then runs three times on those identical inputs -- HIP kernel
(C ABI, the 1:1 entry points), CPU restatement, double precision -- and the two fp16 results are held against the exact one.
tools/op_accuracy_real.py [model] [position] [layers, comma separated]"""
import ctypes as C
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama_cu_awq_amd import api as q4, synth   # noqa: E402
import oracle as orc                             # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "7b"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 24
layers = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "2,15,31").split(",")]
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
dim, hidden, nl, heads, kvh, vocab, seq_len, theta = struct.unpack("<7if", open(path, "rb").read(32))
assert heads == kvh
hs = dim // heads
blob = np.memmap(path, dtype=np.uint8, mode="r")


def qw_at(off, K, N):
    a, b, c = synth.qweight_sizes(K, N)
    w = np.frombuffer(blob, np.uint32, a, off); off += 4 * a
    z = np.frombuffer(blob, np.uint32, b, off); off += 4 * b
    s = np.frombuffer(blob, np.float16, c, off); off += 2 * c
    return (np.array(w), np.array(z), np.array(s)), off


def layer_tensors(l):
    per = 0
    for (h_, w_) in ((dim, dim), (dim, dim), (dim, dim), (dim, dim), (dim, hidden), (dim, hidden), (hidden, dim)):
        a, b, c = synth.qweight_sizes(h_, w_)
        per += a * 4 + b * 4 + c * 2
    per += 2 * dim * 2
    off = 32 + 2 * vocab * dim * 2 + dim * 2 + l * per
    t = {}
    for name, K, N in (("q", dim, dim), ("k", dim, dim), ("v", dim, dim), ("o", dim, dim), ("up", dim, hidden), ("gate", dim, hidden), ("down", hidden, dim)):
        t[name], off = qw_at(off, K, N)
    t["rms_att"] = np.array(np.frombuffer(blob, np.float16, dim, off)); off += 2 * dim
    t["rms_ffn"] = np.array(np.frombuffer(blob, np.float16, dim, off))
    return t


m = orc.Model(path)
d16, _ = m.layer_dump()
toks = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
xin = {}
for pos in range(P + 1):
    lg = m.forward(toks[pos], pos)
    if pos + 1 >= len(toks):
        toks.append(int(np.argmax(lg.astype(np.float32))))
    if pos == P:
        emb = np.array(np.frombuffer(blob, np.float16, dim, 32 + toks[pos] * dim * 2))
        for l in layers:
            xin[l] = emb if l == 0 else d16[l - 1, 1].copy()
kc_all, vc_all = m.kv()
m.close()

L = q4.lib()
q4.check(L.q4_set_device(0))
s = C.c_void_p()
q4.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)


def report(name, gpu, rest, exact):
    exact = np.asarray(exact, np.float64).ravel()
    g, r = np.asarray(gpu, np.float64).ravel(), np.asarray(rest, np.float64).ravel()
    ag, ar = np.sqrt(np.mean((g - exact) ** 2)), np.sqrt(np.mean((r - exact) ** 2))
    print("    %-30s abs rms err HIP %.4e  restatement %.4e  ratio %.4f   | max HIP %.3e rest %.3e | rms of exact %.3f | outputs that differ %.4f" % (
        name, ag, ar, ag / ar, np.abs(g - exact).max(), np.abs(r - exact).max(), np.sqrt(np.mean(exact ** 2)), np.mean(g != r)), flush=True)


def gemv(name, xv, t, K, N, old=None):
    w, z, sc = t
    ex = orc.matmul_q4_f64(xv, w, z, sc, K, N) + (old.astype(np.float64) if old is not None else 0.0)
    rest = orc.matmul_q4(xv, w, z, sc, K, N, accum_into=old)
    dw = q4.DevQWeight(w, z, sc)
    dx, dout = q4.DevBuf(xv), q4.DevBuf(old if old is not None else np.zeros(N, np.float16))
    q4.matmul_q4(dout, dx, dw, K, N, accum=old is not None)
    q4.synchronize()
    g = dout.get(np.float16, N)
    report(name, g, rest, ex)
    return rest


for l in layers:
    t = layer_tensors(l)
    x = xin[l]
    print("layer %d, position %d (x rms %.3f)" % (l, P, np.sqrt(np.mean(x.astype(np.float64) ** 2))))
    x64 = x.astype(np.float64)
    ex = x64 * (t["rms_att"].astype(np.float64) / np.sqrt(np.mean(x64 ** 2) + 1e-5))
    dx, dw, do = q4.DevBuf(x), q4.DevBuf(t["rms_att"]), q4.DevBuf(nbytes=dim * 2)
    q4.rmsnorm(do, dx, dw, dim); q4.synchronize()
    xb = orc.rmsnorm(x, t["rms_att"])
    report("rmsnorm (attention)", do.get(np.float16, dim), xb, ex)
    qv = gemv("q GEMV", xb, t["q"], dim, dim)
    kv = gemv("k GEMV", xb, t["k"], dim, dim)
    vv = gemv("v GEMV", xb, t["v"], dim, dim)
    i = np.arange(hs // 2)
    freq = 1.0 / theta ** (2.0 * i / hs)
    c, sn = np.cos(P * freq), np.sin(P * freq)

    def rope64(v):
        v = v.astype(np.float64).reshape(heads, hs)
        a, b = v[:, : hs // 2], v[:, hs // 2:]
        return np.concatenate([a * c - b * sn, a * sn + b * c], axis=1).ravel()
    kcache = np.zeros((P + 1, dim), np.float16)        # the kernel rotates the key row INSIDE the cache, at row P (gpu_kernels.h:347-354)
    kcache[P] = kv
    dq, dk, dpos = q4.DevBuf(qv), q4.DevBuf(kcache), q4.DevBuf(np.array([P], np.int32))
    q4.RoPERotation(dq, dk, heads, heads, hs, dpos, 0, theta); q4.synchronize()
    rq, rk = orc.rope(qv, kv, heads, heads, hs, P, theta)
    report("rope (q)", dq.get(np.float16, dim), rq, rope64(qv))
    report("rope (k, in the cache)", dk.get(np.float16).reshape(P + 1, dim)[P], rk, rope64(kv))
    kc, vc = kc_all[l, : P + 1].copy(), vc_all[l, : P + 1].copy()        # the restatement's own cache rows (position P = rk, vv)
    q64 = rq.astype(np.float64).reshape(heads, hs)
    k64, v64 = kc.astype(np.float64).reshape(P + 1, heads, hs), vc.astype(np.float64).reshape(P + 1, heads, hs)
    sc64 = np.einsum("hd,thd->ht", q64, k64) / np.sqrt(hs)
    p64 = np.exp(sc64 - sc64.max(axis=1, keepdims=True)); p64 /= p64.sum(axis=1, keepdims=True)
    ex = np.einsum("ht,thd->hd", p64, v64).ravel()
    seq = 128
    kcp, vcp = np.zeros((seq, dim), np.float16), np.zeros((seq, dim), np.float16)
    kcp[: P + 1], vcp[: P + 1] = kc, vc
    dq2, dkc, dvc, dout = q4.DevBuf(rq), q4.DevBuf(kcp), q4.DevBuf(vcp), q4.DevBuf(nbytes=dim * 2)
    datt = q4.DevBuf(nbytes=heads * 8 * (hs + 4) * 8)
    q4.MultiHeadAttention(dout, dq2, dkc, dvc, datt, heads, hs, 1, seq, dpos); q4.synchronize()
    att, _ = orc.attention(rq, kcp, vcp, heads, hs, 1, P, seq)
    report("attention", dout.get(np.float16, dim), att, ex)
    x1 = gemv("o-proj GEMV + residual", att, t["o"], dim, dim, old=x)
    x164 = x1.astype(np.float64)
    ex = x164 * (t["rms_ffn"].astype(np.float64) / np.sqrt(np.mean(x164 ** 2) + 1e-5))
    dx, dw = q4.DevBuf(x1), q4.DevBuf(t["rms_ffn"])
    q4.rmsnorm(do, dx, dw, dim); q4.synchronize()
    xb2 = orc.rmsnorm(x1, t["rms_ffn"])
    report("rmsnorm (ffn)", do.get(np.float16, dim), xb2, ex)
    g64, u64 = orc.matmul_q4_f64(xb2, *t["gate"], dim, hidden), orc.matmul_q4_f64(xb2, *t["up"], dim, hidden)
    ex = g64 / (1.0 + np.exp(-g64)) * u64
    dg, du = q4.DevQWeight(*t["gate"]), q4.DevQWeight(*t["up"])
    dxb, dh = q4.DevBuf(xb2), q4.DevBuf(nbytes=hidden * 2)
    q4.ffn_matvec_silu(dh, dxb, dg, du, dim, hidden); q4.synchronize()
    hb = orc.ffn_matvec_silu(xb2, t["gate"], t["up"], dim, hidden)
    report("gate/up + SiLU", dh.get(np.float16, hidden), hb, ex)
    gemv("down GEMV + residual", hb, t["down"], hidden, dim, old=x1)
