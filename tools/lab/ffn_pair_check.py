#!/usr/bin/env python3
"""The FFN half of a layer as one launch (csrc/gemv_ffn_pair.h, fusion level 4) against the launch sequence (level 3), in one process:
bit equality of logits, K / V rows and greedy token rings; interleaved tokens/s; and -- with the profiling build -- the launch's wall-clock
timeline.   tools/lab/ffn_pair_check.py [model] [ntok] [pre]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth
api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pres = [int(v.rstrip("n")) + (1000 if v.endswith("n") else 0) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]   # 8n: pre 8 with nt gather loads
pre = pres[0]
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
t = api.Transformer(path)
L.q4_set_gemv_early(16, pre % 1000); L.q4_set_gemv_early(18, pre // 1000)
prompt = [1, 2436, 385, 3686, 388, 1048, 22796, 118]
out = {}
LEVELS = tuple(int(v) for v in os.environ.get("LEVELS", "3,4,5,6").split(","))
for lvl in LEVELS:
    L.q4_set_fusion(lvl)
    for graphs in (0, 1):
        L.q4_set_use_graphs(graphs)
        t.reset(prompt)
        logits, kv = [], []
        for pos in range(12):
            t.run_transformer(pos >= len(prompt) - 1, False)
            api.synchronize()
            logits.append(t.logits().view(np.uint16).copy())
            kv.append(np.concatenate([np.concatenate(t.kv_row(l, pos)) for l in (0, 1, t.config.n_layers - 1)]).view(np.uint16).copy())
        out[(lvl, graphs)] = (np.stack(logits), np.stack(kv))
    L.q4_set_use_graphs(1)
    out[(lvl, "ring")] = t.generate_ids(prompt, 64)[0].copy()
    print("level %d: fusion now %d, hand-off time-outs %d" % (lvl, L.q4_get_fusion(), L.q4_handoff_timeouts()), flush=True)
ok = True
for lv in LEVELS[1:]:
  for g in (0, 1):
    for k, name in ((0, "logits"), (1, "kv rows")):
        same = np.array_equal(out[(3, g)][k], out[(lv, g)][k])
        ok &= same
        print("graphs %d %-8s level %d == level 3: %s" % (g, name, lv, same), flush=True)
        if not same:
            d = np.argwhere(out[(3, g)][k] != out[(lv, g)][k])
            print("   first differences (pos, index):", d[:8].tolist(), " count", len(d))
  same = np.array_equal(out[(3, "ring")], out[(lv, "ring")])
  ok &= same
  print("token rings equal (level %d): %s" % (lv, same), flush=True)
print("finite logits:", bool(np.isfinite(out[(LEVELS[-1], 1)][0].view(np.float16).astype(np.float32)).all()))

# speed, interleaved
for pre in pres:
  L.q4_set_gemv_early(16, pre % 1000); L.q4_set_gemv_early(18, pre // 1000)
  res = {l: [] for l in LEVELS}
  for rep in range(3):
    for lvl in LEVELS:
        L.q4_set_fusion(lvl)
        t.generate_ids(prompt, ntok)
        res[lvl].append(max(t.generate_ids(prompt, ntok)[1] for _ in range(3)))
  for lvl in LEVELS:
    print("per token: level 3 %.4f ms, level %d %.4f ms, difference %.1f us" % (1e3 / np.median(res[3]), lvl, 1e3 / np.median(res[lvl]), 1e6 / np.median(res[3]) - 1e6 / np.median(res[lvl]))) if lvl != 3 else None
    print("pre %d level %d  -n %d  best-of-3 tokens/s per round: %s   median %.1f" % (pre, lvl, ntok, " ".join("%.1f" % v for v in res[lvl]), float(np.median(res[lvl]))), flush=True)
  print("time-outs:", L.q4_handoff_timeouts(), "fusion:", L.q4_get_fusion())

# timeline of one launch (eager, profiling build): the stamps of the last layer's launch
L.q4_set_fusion(int(os.environ.get("TIMELINE_LEVEL", str(LEVELS[-1]))))
L.q4_set_use_graphs(0)
nb = 256
dbg = api.DevBuf(nbytes=4 << 20)
t.reset(prompt)
for pos in range(4):
    t.run_transformer(pos >= len(prompt) - 1, False)
api.synchronize()
for trial, pre in enumerate(pres):
    L.q4_set_gemv_early(16, pre % 1000); L.q4_set_gemv_early(18, pre // 1000)
    L.q4_set_debug_buffer(dbg.ptr)
    t.run_transformer(False, False)
    api.synchronize()
    L.q4_set_debug_buffer(None)
    raw = dbg.get(np.uint64)
    st = raw[: nb * 64].reshape(nb, 64).astype(np.int64)
    s2 = raw[nb * 64: nb * 128].reshape(nb, 64).astype(np.int64)
    good = (st[:, 11] > 0) & (st[:, 3] > 0)
    print("blocks with a complete record: %d of %d" % (good.sum(), nb))
    st, s2 = st[good], s2[good]
    dd = s2[:, 0:16]
    t0 = st[:, 0].min()
    us = lambda v: (v - t0) * 0.01

    def row(name, v):
        v = us(np.asarray(v)).ravel()
        print("%-46s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))

    print("--- timeline, pre %d (last layer's launch; %d blocks)" % (pre, nb))
    row("wave 0 entry", st[:, 0])
    if (s2[128:, 41] > 0).all():      # the whole-layer launch (fusion level 6): blocks [0, 128) attention units, [128, 256) the output projection
        na = 128
        row("attention role done (blocks with a unit)", s2[:na, 40]); row("o-proj blocks: requests out", s2[na:, 40])
        row("o-proj blocks: attention output gathered, staged", s2[na:, 41]); row("o-proj blocks: x published (wave 0)", s2[na:, 42])
        row("x gathered (wave 1)", s2[:, 43]); row("   ... blocks with a unit", s2[:na, 43]); row("   ... o-proj blocks", s2[na:, 43])
        row("x staged: blocks with a unit", st[:na, 3]); row("x staged: o-proj blocks", st[na:, 3])
        row("gate/up done (last wave): blocks with a unit", st[:na, 32:48].max(axis=1)); row("gate/up done (last wave): o-proj blocks", st[na:, 32:48].max(axis=1))
    row("x landed", st[:, 1]); row("sum of squares exchanged", st[:, 2]); row("x staged", st[:, 3])
    row("gate/up done, per wave", st[:, 32:48]); row("gate/up done, block's last wave", st[:, 32:48].max(axis=1))
    row("barrier A passed (wave 0)", st[:, 4]); row("published (stores issued)", st[:, 5])
    row("wave 1: first gather pass issued", st[:, 8]); row("wave 1: first gather pass back", st[:, 6]); row("wave 1: gathered", st[:, 7])
    row("gathered, per wave (1..15)", st[:, 17:32]); row("gathered, block's last wave", st[:, 17:32].max(axis=1))
    row("down pieces landed, per wave", st[:, 48:64]); row("down pieces landed, block's last wave", st[:, 48:64].max(axis=1))
    row("barrier B passed (wave 1)", st[:, 9]); row("wave 1: dots done", st[:, 10])
    row("phase 2 dots done, per wave", dd); row("phase 2 dots done, block's last wave", dd.max(axis=1))
    if (s2[:, 33] > 0).all():      # the three-phase launch (fusion level 5)
        row("barrier C passed (wave 0)", s2[:, 32]); row("x published (wave 0)", s2[:, 33]); row("wave 1: first x pass back", s2[:, 34]); row("wave 1: x gathered", s2[:, 35])
        row("sum of squares exchanged (wave 1)", s2[:, 36]); row("x staged (wave 1)", s2[:, 37])
        row("phase 3 dots done, per wave", s2[:, 16:32]); row("phase 3 dots done, block's last wave", s2[:, 16:32].max(axis=1)); row("q / k / v stored (wave 0)", s2[:, 38])
        print("x gather passes of wave 1: min %d median %d max %d" % (s2[:, 39].min(), np.median(s2[:, 39]), s2[:, 39].max()))
    row("stored (wave 0)", st[:, 11])
    print("gather passes of wave 1: min %d median %d max %d" % (st[:, 12].min(), np.median(st[:, 12]), st[:, 12].max()))
    last_pub = us(st[:, 5]).max()
    print("last publish %.2f us; launch ends %.2f us; last publish -> end %.2f us" % (last_pub, us(st[:, 11]).max(), us(st[:, 11]).max() - last_pub))
L.q4_set_use_graphs(1)
t.close()
print("RESULT", "PASS" if ok else "FAIL")
