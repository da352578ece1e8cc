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



def qweight_bytes(K, N):
    return ((K + 31) // 32 * 4) * N * 4 + (((K + 127) // 128 + 7) // 8) * N * 4 + ((K + 127) // 128) * N * 2


GEOM = {"7b": (4096, 11008, 32000, 32), "13b": (5120, 13824, 32000, 40)}


def algorithmic(kernel, model, ntok):
    """algorithmic bytes per launch of a kernel of the decode network (weights + metadata + vectors in / out)"""
    d, h, v, _ = GEOM[model]
    kernel = kernel[4:] if kernel.startswith("q4::") else kernel
    if kernel.startswith("gemv_q4_kernel<2") or kernel.startswith("ffn_engine_kernel") or kernel.startswith("ffn_strip_kernel") or kernel.startswith("ffn_strip_pair_kernel"):
        return 2 * qweight_bytes(d, h) + 2 * d * 2 + h * 2
    if kernel.startswith("ffn_pair_kernel"):     # fusion level 4: gate/up + down in one launch (bench.py kernel_bytes id 10); the hb hand-off inside the launch is not algorithmic
        pair = 2 * qweight_bytes(d, h) + qweight_bytes(h, d) + 4 * d * 2 + h * 2
        if re.match(r"ffn_pair_kernel<\w+, \w+, true[,>]", kernel):   # fusion level 5: + the next layer's rmsnorm + q/k/v + RoPE + KV write (id 11)
            return pair + 3 * qweight_bytes(d, d) + d * 2 + 3 * d * 2
        return pair
    if kernel.startswith("gemv_q4_kernel<1"):
        return 3 * qweight_bytes(d, d) + 2 * d * 2 + 3 * d * 2
    if kernel.startswith("gemv_q4_kernel<0") or kernel.startswith("down_strip_kernel"):
        return qweight_bytes(h, d) + h * 2 + 2 * d * 2
    if kernel.startswith("gemv_f16_kernel") or kernel.startswith("cls_strip_kernel"):
        return v * d * 2 + d * 2 + v * 2
    if kernel.startswith("attention_oproj_kernel") or kernel.startswith("attention_oproj16_kernel"):      # (16: the sixteen-wave launch of forms 5 / 6)
        # o-proj QWeight + residual in / out + q + attention output + K and V rows of the context, averaged over the context
        # lengths this form serves in a -n ntok run launched with the graph path's BINS (q4_set_use_graphs(2): tools/prof_decode.py,
        # bench.py --no-graphs; round 5 -- up to round 4 the eager run passed the exact context length and form 1 served 257..511).
        # form = third template argument (layer_attn.hip, ao_shape): 5 = V-slice blocks, bin 128; 6 = V-slice blocks, bin 256
        # (positions 129..256); 4 = 64-position chunks, bin 512 (257..512); 2 = 128-position chunks, bin 1024; 3 = 256-position chunks
        # above; 0 / 1 = one block per head (V-slice form switched off; not what the product runs)
        m = re.match(r"attention_oproj_kernel<\d+, \w+, (\d)", kernel) or re.match(r"attention_oproj16_kernel<(\d)", kernel)
        form = int(m.group(1)) if m else -1
        ranges = {0: (1, 128), 5: (1, 128), 1: (129, 256), 6: (129, 256), 4: (257, 512), 2: (513, 1024), 3: (1025, 2048)}
        if form not in ranges:
            raise SystemExit("attention_oproj_kernel form %d has no context range in summarize_profiles.py: %s" % (form, kernel))
        lo, hi = ranges[form]
        hi = min(hi, ntok)
        avg_rows = (lo + hi) / 2.0
        return int(qweight_bytes(d, d) + 4 * d * 2 + avg_rows * 2 * d * 2)
    return None


traffic = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*_FETCH_SIZE"))):
    m = re.match(r"pmc_(\w+)_(\d+)_FETCH_SIZE", os.path.basename(d))
    if not m or not os.path.isdir(d):
        continue
    model, ntok = m.group(1), int(m.group(2))
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = defaultdict(list)
        for f in glob.glob(os.path.join(src, "pmc_%s_%d_%s" % (model, ntok, c), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            per.setdefault(k, {})[c + "_KB_avg"] = round(sum(v) / len(v), 2)
            per[k]["dispatches"] = len(v)
    for k, e in per.items():
        e["traffic_bytes_per_launch"] = int(2 * 1024 * e.get("FETCH_SIZE_KB_avg", 0) + 1024 * e.get("WRITE_SIZE_KB_avg", 0))
        alg = algorithmic(k, model, ntok)
        if alg:
            e["algorithmic_bytes_per_launch"] = alg
            e["traffic_over_algorithmic"] = round(e["traffic_bytes_per_launch"] / alg, 4)
            # a launch cannot move fewer bytes than its algorithm needs: a ratio below 1 means the byte model above is wrong
            assert e["traffic_over_algorithmic"] >= 0.98, (k, model, ntok, e)
    traffic["%s_n%d" % (model, ntok)] = per
    print(model, ntok, json.dumps(per)[:600])
if "7b_n256" in traffic:     # bench.py reads the dominant kernel's figure from here
    for k, e in traffic["7b_n256"].items():
        if k.replace("q4::", "").startswith(("gemv_q4_kernel<2", "ffn_engine_kernel", "ffn_strip_kernel", "ffn_strip_pair_kernel")) and "0" not in traffic:   # the fused gate/up launch in whatever form the product runs it
            traffic["0"] = {"kernel": "ffn_rmsnorm_gate_up_silu_q4", "traffic_bytes_per_launch": e["traffic_bytes_per_launch"],
                            "algorithmic_bytes_per_launch": e.get("algorithmic_bytes_per_launch")}
        three_phase = any(kk.replace("q4::", "").startswith("ffn_pair_kernel<true, false, true") for kk in traffic["7b_n256"])
        if k.replace("q4::", "").startswith("ffn_pair_kernel<true, false, true" if three_phase else "ffn_pair_kernel"):      # ... or, at fusion levels 4 / 5, the FFN launch: the decode path's dominant launch
            traffic["0"] = {"kernel": "ffn_gate_up_down_accum_then_next_qkv_rope_q4" if three_phase else "ffn_rmsnorm_gate_up_silu_down_accum_q4", "traffic_bytes_per_launch": e["traffic_bytes_per_launch"],
                            "algorithmic_bytes_per_launch": e.get("algorithmic_bytes_per_launch")}
traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh) over eager greedy decodes of the "
                   "product's own launch sequence (tools/prof_decode.py). FETCH_SIZE on gfx950 tallies the 128-B requests of 16 B/lane "
                   "streaming reads at 64 B (MI355X_MICROARCH.md, HBM section): doubled before comparing with byte counts. Kernels that read "
                   "narrower than 16 B per lane (attention merges, argmax) are outside that calibration.")
json.dump(traffic, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)


# SQ counters of the gate/up launch run alone (tools/prof_kernel.py 0): VALU instruction count and VALU-busy time per launch -- the
# evidence behind "the int4 GEMV is VALU-bound" (EXPERIMENTS.md notebook §9.13)
sq = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(src, "pmc_k0_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(n in r["Kernel_Name"] for n in ("gemv_q4_kernel<2", "ffn_engine_kernel", "ffn_strip_kernel", "ffn_strip_pair_kernel", "ffn_pair_kernel")):
            sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"kernel": short(r["Kernel_Name"]), "vgpr": int(r["VGPR_Count"]), "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
if sq:
    out = {k: round(sum(v) / len(v), 1) for k, v in sq.items()}
    out.update(meta)
    out["launches"] = len(next(iter(sq.values())))
    if "SQ_ACTIVE_INST_VALU" in out and "SQ_INSTS_VALU" in out:
        simds = 1024
        out["valu_busy_cycles_per_simd"] = round(4.0 * out["SQ_ACTIVE_INST_VALU"] / simds, 1)   # the counter ticks in quad-cycles
        out["cycles_per_valu_instruction"] = round(4.0 * out["SQ_ACTIVE_INST_VALU"] / out["SQ_INSTS_VALU"], 2)
        out["valu_instructions_per_wave"] = round(out["SQ_INSTS_VALU"] / max(1.0, out.get("SQ_WAVES", 1.0)), 1)
    out["note"] = ("rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES over "
                   "tools/prof_kernel.py <id> 32 (id 11: the FFN half + the next layer's QKV as one launch, fusion level 5; id 10: the FFN pair launch alone, level 4; id 0: the fused gate/up launch -- over the ring of the layers' weights); SQ_ACTIVE_INST_VALU counts quad-cycles "
                   "summed over the chip's 1024 SIMDs (MI355X_MICROARCH.md, per-instruction constants)")
    # (the FFN pair launch, where the round's collection timed kernel 10: the decode path's dominant launch at fusion level 4)
    json.dump(out, open(os.path.join(dst, "%s_sq_counters_%s.json" % (tag, "ffn_pair" if "ffn_pair_kernel" in out.get("kernel", "") else "gate_up")), "w"), indent=1)
    print("SQ counters:", json.dumps(out)[:400])
