#!/usr/bin/env python3
"""Wall-clock timeline (s_memrealtime, 10 ns) of the fused attention-block launch, per role: when the QKV blocks end,
when the attention blocks pass their wait and end, when the o-proj blocks pass theirs and end. Profiling build.
tools/lab/timeline_block.py [model] [position] [fusion] [chunk min_bin]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llama_cu_awq_amd import api, synth   # noqa: E402

api.use_profiling_build()
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
upto = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fusion = int(sys.argv[3]) if len(sys.argv) > 3 else 3
path = "/tmp/llama2_q4_synth_%s_seed20240229.bin" % model
if not os.path.exists(path):
    synth.write_model(path, model)
L = api.lib()
api.check(L.q4_set_device(0))
s = C.c_void_p()
api.check(L.q4_stream_create(C.byref(s)))
L.q4_set_stream(s)
L.q4_set_fusion(fusion)
if len(sys.argv) > 5:      # split-context setting: positions per block, smallest bin that splits
    L.q4_set_attention_split(int(sys.argv[4]), int(sys.argv[5]))
for kv in os.environ.get("KNOBS", "").split(","):      # profiling knobs, e.g. KNOBS=14=0 (the split-context role's K / V rows in registers)
    if kv:
        L.q4_set_gemv_early(int(kv.split("=")[0]), int(kv.split("=")[1]))
tr = api.Transformer(path)
tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], upto)
L.q4_set_use_graphs(0)
nblocks = 4096
dbg = api.DevBuf(nbytes=nblocks * 4 * 8)
for rep in range(3):
    L.q4_set_debug_buffer(dbg.ptr)
    tr.run_transformer(True)
    api.synchronize()
    L.q4_set_debug_buffer(None)
    d = dbg.get(np.uint64).reshape(nblocks, 4).astype(np.int64)
    used = d[:, 0] > 0
    d = d[used]
    t0 = d[:, 0].min()
    print("rep %d: %d blocks, launch span %.2f us" % (rep, len(d), (d[:, 2].max() - t0) / 100.0))
    role_of = d[:, 3] & 0xFF
    hw = (d[:, 3] >> 8) & 0xFFFF
    xcc = (d[:, 3] >> 32) & 0xF
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))      # xcc, se, sh, cu
    if rep == 2 and fusion == 2:
        q = d[role_of == 0]
        order = np.argsort(q[:, 2])
        qcu = cu[role_of == 0]
        others = set(cu[role_of != 0].tolist())
        att_cus = set(cu[role_of == 1].tolist())
        print("  qkv end percentiles (us):", [round(float(v), 2) for v in np.percentile((q[:, 2] - t0) / 100.0, [10, 50, 75, 90, 95, 99, 100])])
        late = (q[:, 2] - t0) / 100.0 > 8.0
        print("  late qkv blocks: %d; of them on a CU that also hosts an attention block: %d, an o-proj/attention block: %d; qkv blocks on shared CUs overall: %d" % (
            late.sum(), sum(1 for c in qcu[late] if c in att_cus), sum(1 for c in qcu[late] if c in others), sum(1 for c in qcu if c in others)))
        ids = np.nonzero(role_of == 0)[0]
        lt = (q[:, 2] - t0) / 100.0
        print("  late block ids:", ids[late].tolist())
        print("  end time by matrix (q,k,v) median:", [round(float(np.median(lt[(ids // 128) == m])), 2) for m in range(3)])
        # per CU: how many qkv blocks, and end times
        from collections import defaultdict
        percu = defaultdict(list)
        for i, c in zip(ids, qcu):
            percu[int(c)].append(round(float(lt[list(ids).index(i)]), 1))
        two = [v for v in percu.values() if len(v) == 2]
        one = [v for v in percu.values() if len(v) == 1]
        print("  CUs with one qkv block: %d, end median %.2f max %.2f; with two: %d, first-ending median %.2f, second-ending median %.2f" % (
            len(one), np.median([v[0] for v in one]) if one else 0, max([v[0] for v in one]) if one else 0, len(two),
            np.median([min(v) for v in two]) if two else 0, np.median([max(v) for v in two]) if two else 0))
        print("  distinct CUs: qkv %d, attention %d, o-proj %d" % (len(set(qcu.tolist())), len(att_cus), len(set(cu[role_of == 2].tolist()))))
    for role, name in ((0, "producer"), (1, "attention"), (2, "consumer")):
        r = d[role_of == role]
        if not len(r):
            continue
        e, w, x = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0
        line = "  %-9s %4d blocks  entry %.2f..%.2f" % (name, len(r), e.min(), e.max())
        if role == 0:
            line += "   math done %.2f / %.2f / %.2f" % (w.min(), np.median(w), w.max())
        if role:
            line += "   passed wait %.2f / %.2f / %.2f (min/med/max)" % (w.min(), np.median(w), w.max())
        line += "   end %.2f / %.2f / %.2f" % (x.min(), np.median(x), x.max())
        print(line)
tr.close()
