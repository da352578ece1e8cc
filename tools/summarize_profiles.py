#!/usr/bin/env python3
"""Boil the rocprofv3 outputs of tools/profile_round.sh down to the small files committed under profiles/:
  <tag>_kernel_stats_<cfg>.csv  per-kernel calls / total / average / min / max duration from the kernel trace
  <tag>_traffic.json            per timed kernel: FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE per
                                launch next to the algorithmic bytes, and the trace's average duration
usage: summarize_profiles.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "summaries")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").strip()


def trace_stats(d):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return rows


for d in sorted(glob.glob(os.path.join(src, "trace_*"))):
    if not os.path.isdir(d):
        continue
    rows = trace_stats(d)
    if not rows:
        print("no kernel trace in", d)
        continue
    cfg = os.path.basename(d)[len("trace_"):]
    total = sum(sum(v) for v in rows.values())
    with open(os.path.join(dst, "%s_kernel_stats_%s_eager.csv" % (tag, cfg)), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v), 1), round(sum(v) / len(v), 3), round(min(v), 3), round(max(v), 3), round(100 * sum(v) / total, 2)])
    print(cfg, "kernels:", len(rows), "total ms", round(total / 1e3, 1))
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("   %-90s %6d calls avg %8.3f us" % (k[:90], len(v), sum(v) / len(v)))

KERNELS = {0: "gemv_q4_kernel<2", 2: "gemv_q4_kernel<0, 3", 3: "gemv_q4_kernel<1", 4: "gemv_q4_kernel<0, 2", 5: "gemv_f16_kernel", 6: "attention"}
NAMES = {0: "ffn_rmsnorm_gate_up_silu_q4", 2: "gemv_q4_hidden_to_dim_accum (down)", 3: "qkv_rmsnorm_rope_q4", 4: "gemv_q4_oproj_accum", 5: "classifier_f16",
         6: "attention, stand-alone one-block kernel (q4_multi_head_attention)"}


def qweight_bytes(K, N):
    return ((K + 31) // 32 * 4) * N * 4 + (((K + 127) // 128 + 7) // 8) * N * 4 + ((K + 127) // 128) * N * 2


d_, h_, v_ = 4096, 11008, 32000
ALG = {0: 2 * qweight_bytes(d_, h_) + 2 * d_ * 2 + h_ * 2, 2: qweight_bytes(h_, d_) + h_ * 2 + 2 * d_ * 2, 3: 3 * qweight_bytes(d_, d_) + 2 * d_ * 2 + 3 * d_ * 2,
       4: qweight_bytes(d_, d_) + d_ * 2 + 2 * d_ * 2, 5: v_ * d_ * 2 + d_ * 2 + v_ * 2, 6: None}
traffic = {}
for kid, pat in KERNELS.items():
    ent = {"kernel": NAMES[kid], "algorithmic_bytes_per_launch": ALG[kid]}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = defaultdict(list)
        for f in glob.glob(os.path.join(src, "pmc_k%d_%s" % (kid, c), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and pat in r["Kernel_Name"]:
                    vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            ent.setdefault("per_kernel", {}).setdefault(k, {})[c + "_KB_avg"] = round(sum(v) / len(v), 2)
            ent["per_kernel"][k]["dispatches"] = len(v)
    tot = 0.0
    for k, e in ent.get("per_kernel", {}).items():
        e["traffic_bytes_per_launch"] = int(2 * 1024 * e.get("FETCH_SIZE_KB_avg", 0) + 1024 * e.get("WRITE_SIZE_KB_avg", 0))
        tot += e["traffic_bytes_per_launch"]
    if ent.get("per_kernel"):
        ent["traffic_bytes_per_launch"] = int(tot)
        if ALG[kid]:
            ent["traffic_over_algorithmic"] = round(tot / ALG[kid], 4)
    traffic[str(kid)] = ent
    print(kid, json.dumps(ent)[:300])
traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), 32 dispatches over the 32 layers' "
                   "weights each (tools/prof_kernel.py <id> 32, 7B geometry). FETCH_SIZE on gfx950 tallies the 128-B requests of 16 B/lane "
                   "streaming reads at 64 B (MI355X_MICROARCH.md, HBM section): doubled before comparing with byte counts.")
json.dump(traffic, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
