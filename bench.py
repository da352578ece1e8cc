#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X-native llama2_q4 on Llama-2-7B-geometry AWQ int4 g128 weights.

Contract (one JSON line on rank 0):
  step     = one full greedy `-n 256` generation (8-token prompt fed one token per step + 247 generated tokens =
             255 timed tokens, the reference's own rule (pos-1)/elapsed, llama2_q4.cu:485-489), weights resident
             in HBM before the timed region.
  value    = total timed tokens of all ranks / max-over-ranks wall time of K steps  (whole-job tokens/s).
  N > 1    = N independent replicas (the token loop is sequential, SURVEY 8e: replicas only, no data-path
             collective); torch.distributed is used for the barrier and the max-over-ranks only.
  roofline = the decode path's dominant kernel (fused rmsnorm + gate/up int4 GEMV + SiLU): algorithmic bytes per
             launch / its average launch duration, measured here with HIP events (dispatch timestamps on the launch
             stream) on every gate/up launch of 16 eager decode steps of the same network (in context: x comes
             from the o-proj kernel, 3.6 GB of weights stream between two uses), against the 8 TB/s HBM3E peak.
             `kernels` also lists each GEMV alone over a ring of the 32 layers' weights.
  cpu_baseline = the CPU restatement (oracle/, "port") of run_llama_network timed on the host cores for a few
             decode steps of the SAME checkpoint; it is a reported baseline, not the target.
Synthetic weights: no network, no real checkpoint -- llama_cu_awq_amd/synth.py, seed 20240229.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROMPT_IDS = [1, 2436, 385, 3686, 388, 1048, 22796, 118]   # encode("write an essay about GPUs", bos=1), reference tokenizer
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PUBLISHED_TOKS = 200.787402                                 # README.md:111, RTX 4090


def qweight_bytes(K, N):
    pwh = (K + 31) // 32 * 4
    sh = (K + 127) // 128
    pzh = (sh + 7) // 8
    return pwh * N * 4 + pzh * N * 4 + sh * N * 2


def kernel_bytes(cfg):
    """Algorithmic bytes per launch of each timed kernel (weights + metadata + vectors in/out)."""
    d, h, v = cfg.dim, cfg.hidden_dim, cfg.vocab_size
    return {
        0: ("ffn_rmsnorm_gate_up_silu_q4", 2 * qweight_bytes(d, h) + 2 * d * 2 + h * 2),
        1: ("gemv_q4_dim_to_hidden", qweight_bytes(d, h) + d * 2 + h * 2),
        2: ("gemv_q4_hidden_to_dim_accum", qweight_bytes(h, d) + h * 2 + 2 * d * 2),
        3: ("qkv_rmsnorm_rope_q4", 3 * qweight_bytes(d, d) + 2 * d * 2 + 3 * d * 2),
        4: ("gemv_q4_oproj_accum", qweight_bytes(d, d) + d * 2 + 2 * d * 2),
        5: ("classifier_f16", v * d * 2 + d * 2 + v * 2),
        # fusion level 4: rmsnorm + gate/up + SiLU + down + residual as ONE launch (csrc/gemv_ffn_pair.h): the three QWeights, x and the norm weights in,
        # the residual in, x out, and RunState::hb, which the launch still leaves behind (its in-launch hand-off of hb is not algorithmic traffic)
        10: ("ffn_rmsnorm_gate_up_silu_down_accum_q4", 2 * qweight_bytes(d, h) + qweight_bytes(h, d) + 4 * d * 2 + h * 2),
        # fusion level 5: ... and rmsnorm + q/k/v + RoPE + KV write of the NEXT layer as the launch's third phase: + three QWeights, the norm weights, q and the two cache rows out
        11: ("ffn_gate_up_down_accum_then_next_qkv_rope_q4", 2 * qweight_bytes(d, h) + qweight_bytes(h, d) + 4 * d * 2 + h * 2 + 3 * qweight_bytes(d, d) + d * 2 + 3 * d * 2),
    }


def sample_sclk_ghz(run, seconds=2.5, device=0):
    """Shader clock of `device` while `run()` keeps the token loop busy: rocm-smi polled from a side thread (the timed region is never sampled: the
    poll costs host time). Returns (GHz, source): the MEDIAN of the samples; (None, reason) when rocm-smi is not usable."""
    import re
    import subprocess
    import threading
    vals, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                mm = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
                if mm:
                    vals.append(int(mm.group(1)))
            except Exception:
                return
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.time()
    while time.time() - t0 < seconds:
        run()
    stop.set()
    th.join(timeout=15)
    vals = [v for v in vals if v > 500]
    if vals:
        vals.sort()
        return vals[len(vals) // 2] / 1000.0, "rocm-smi -d %d --showclocks during the token loop (%d samples, median of %s MHz)" % (device, len(vals), sorted(set(vals)))
    return None, "rocm-smi not usable here"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="7b", help="geometry name in llama_cu_awq_amd.synth.GEOMETRIES")
    ap.add_argument("--ntok", type=int, default=256, help="the CLI's -n")
    ap.add_argument("--model-dir", default=os.environ.get("Q4_MODEL_DIR", "/tmp"))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-steps", type=int, default=32, help="decode steps of the CPU restatement timed for cpu_baseline (~0.4 s each at 7B on 16 cores)")
    ap.add_argument("--f64-steps", type=int, default=10, help="positions of the unrounded double forward used as parity yardstick")
    ap.add_argument("--force-dist", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs leg (13B -n 256, 7B -n 2048)")
    ap.add_argument("--no-graphs", action="store_true", help="eager launches with the graph path's sequence-length bins: what the graphs run, one launch "
                                                               "at a time (rocprofv3 kernel tracing crashes inside hipGraph capture)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched without torch.distributed.run: this process becomes the launcher of --gpus replicas of itself (replicas only,
        # SURVEY 8e; llama2_q4.cu:465-482 is why), rank 0 prints the line. On a box with fewer GPUs q4_set_device fails loudly.
        from llama_cu_awq_amd import replicas
        sys.exit(replicas.spawn(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and (world > 1 or args.gpus > 1):
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE %d: the launcher's world size is what runs\n" % (args.gpus, world))
    use_dist = world > 1 or args.force_dist

    # N > 1: torch (its bundled HIP runtime) must be initialised BEFORE libllama2_q4.so is loaded -- both export
    # libamdhip64.so.7 and the first one loaded serves the whole process; loading ours first leaves torch.cuda
    # with "No HIP GPUs are available". With torch first, the library binds to the runtime torch already loaded.
    dist = None
    dist_device = "cpu"
    if use_dist:
        import torch
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        from llama_cu_awq_amd import replicas
        # a replica without a GPU of its own fails HERE, loudly, before any rendezvous: one rank falling back to another backend while
        # its peers initialise RCCL would leave the whole job hanging in the rendezvous
        if torch.cuda.is_available() and local_rank >= torch.cuda.device_count():
            sys.stderr.write("bench: replica %d of %d has no GPU (this node shows %d): --gpus must not exceed the node's GPUs\n"
                             % (rank, world, torch.cuda.device_count()))
            sys.exit(3)
        try:
            torch.cuda.set_device(local_rank)
            dist = replicas.init("nccl", rank, world)   # nccl == RCCL on ROCm; barrier + two scalar reductions only
            dist_device = "cuda"
            dist.all_reduce(torch.zeros(1, device="cuda"))
            torch.cuda.synchronize()
        except Exception as e:                           # no usable torch.cuda: the replicas still only need a barrier
            sys.stderr.write("bench: RCCL init failed (%s); using gloo for the barrier\n" % e)
            dist = replicas.init("gloo", rank, world)
            dist_device = "cpu"

    from llama_cu_awq_amd import api, synth, replicas   # raises if libllama2_q4.so is missing: no fallback
    L = api.lib()
    api.check(L.q4_set_device(local_rank if world > 1 else 0))
    if args.no_graphs:
        L.q4_set_use_graphs(2)

    def barrier():
        api.check(L.q4_device_synchronize())
        if use_dist:
            dist.barrier()
            if dist_device == "cuda":
                import torch
                torch.cuda.synchronize()

    # ---- synthetic checkpoint (rank 0 writes, everyone loads) --------------------------------------
    geom = synth.GEOMETRIES[args.model]
    path = os.path.join(args.model_dir, "llama2_q4_synth_%s_seed20240229.bin" % args.model)
    t_gen = 0.0
    if rank == 0 and not (os.path.exists(path) and os.path.getsize(path) == synth.model_bytes(geom)):
        t0 = time.time()
        synth.write_model(path, geom)
        t_gen = time.time() - t0
    if use_dist:
        dist.barrier()
    else:
        while not os.path.exists(path):
            time.sleep(0.5)

    stream = C.c_void_p()
    api.check(L.q4_stream_create(C.byref(stream)))
    L.q4_set_stream(stream)
    t0 = time.time()
    tr = api.Transformer(path, temperature=0.0)
    t_load = time.time() - t0
    # plain Python ints copied out at load: nothing below may read the C struct after tr.close() (round 2's record
    # carried garbage config keys read from a freed Transformer)
    class _Cfg:
        pass
    cfg = _Cfg()
    for k_ in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len"):
        setattr(cfg, k_, int(getattr(tr.config, k_)))
    assert (cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size, cfg.seq_len) == tuple(geom[:7]), \
        "loaded header %r differs from geometry %r" % (vars(cfg), geom)
    ntok = min(args.ntok, cfg.seq_len)

    # ---- warmup (also captures the graphs), then K timed generations -------------------------------
    tokens0 = None
    for _ in range(max(args.warmup, 0)):
        tokens0, _, _, _ = tr.generate_ids(PROMPT_IDS, ntok)
    barrier()
    t0 = time.perf_counter()
    timed_tokens = 0
    per_gen = []
    for _ in range(args.steps):
        toks, tps, timed, secs = tr.generate_ids(PROMPT_IDS, ntok)
        timed_tokens += timed
        per_gen.append(tps)
        if tokens0 is None:
            tokens0 = toks
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_tokens = replicas.aggregate(elapsed, timed_tokens, dist if use_dist else None, device=dist_device)
    value = total_tokens / elapsed

    # ---- per-kernel durations (dispatch timestamps, ring over the layers' weights) ------------------
    kb = kernel_bytes(cfg)
    kernels = {}
    roofline = None
    if rank == 0:
        tr.reset(PROMPT_IDS)          # the timed kernels address the KV cache at the device position: back to 0 (after -n 2048 it is seq_len)
        pair = L.q4_get_fusion() >= 4 and L.q4_ffn_pair_covers(cfg.dim, cfg.hidden_dim) == 1    # the FFN half of a layer runs as one launch
        three = pair and L.q4_get_fusion() >= 5 and cfg.n_kv_heads == cfg.n_heads and cfg.dim // cfg.n_heads == 128   # ... with the next layer's QKV inside (csrc/gemv_ffn_pair.hip ffn_qkv_covers)
        dom_id = 11 if three else 10 if pair else 0                                              # the decode path's dominant launch
        for kid, (name, nbytes) in kb.items():
            if (kid == 10 and not pair) or (kid == 11 and not three):
                continue
            tr.bench_kernel(kid, 32)   # warm
            avg, mn, mx = tr.bench_kernel(kid, 256 if kid != 5 else 32)
            kernels[name] = {"us": round(avg, 3), "min_us": round(mn, 3), "bytes": nbytes, "GBps": round(nbytes / avg / 1e3, 1)}
            if kid in (1, 2):   # BASELINE config 1, the single int4 GEMV: back-to-back calls inside one hipGraph over the ring
                g = tr.bench_kernel_graph(kid, 32, 20)
                kernels[name].update({"graph_us_per_call": round(g, 3), "graph_GBps": round(nbytes / g / 1e3, 1)})
        # the roofline kernel's duration is taken where the token loop runs it: eager decode steps of the same
        # network from position 64 on, dispatch timestamps (HIP events on the launch stream) on every gate/up launch
        tr.reset(PROMPT_IDS)
        for pos in range(64):
            tr.run_transformer(pos >= len(PROMPT_IDS) - 1)
            api.synchronize()
        pos_first = tr.pos()
        net_avg, net_min, net_max, net_n = tr.bench_in_network(8, 16)
        graph_us = tr.bench_kernel_graph(dom_id, 32, 20)
        # The roofline is priced on the kernel's own duration. Two measurements exist: HIP events around every gate/up launch of
        # 16 untraced eager decode steps (live, here), and rocprofv3's kernel trace of a whole eager -n 256 decode (committed:
        # profiles/r03_kernel_stats_7b_256_eager.csv, ~6 % longer: every dispatch is intercepted and serialised under the
        # tracer). The line takes the LARGER of the two, so it never claims more than the committed profile supports; the
        # per-launch cost inside a hipGraph (kernel + boundary) is reported next to it.
        rocprof_us, rocprof_src = None, None
        for tag in ("r06", "r05", "r04", "r03", "r02"):
            cpath = os.path.join(ROOT, "profiles", "%s_kernel_stats_%s_%d_eager.csv" % (tag, args.model, ntok))
            if os.path.exists(cpath):
                import csv
                for r in csv.DictReader(open(cpath)):
                    if (("ffn_pair_kernel<true, false, true" in r["kernel"]) if three else ("ffn_pair_kernel<" in r["kernel"])) if pair else ("gemv_q4_kernel<2," in r["kernel"] or "ffn_strip_kernel<" in r["kernel"] or "ffn_strip_pair_kernel<" in r["kernel"]):   # the dominant launch: the FFN pair (csrc/gemv_ffn_pair.h), else the fused gate/up launch (csrc/gemv_strip.h)
                        rocprof_us, rocprof_src = float(r["avg_us"]), "profiles/" + os.path.basename(cpath)
                        break
            if rocprof_us is not None:
                break
        kernel_us = max(net_avg, rocprof_us) if rocprof_us else net_avg
        dom = {"us": round(kernel_us, 3), "GBps": round(kb[dom_id][1] / kernel_us / 1e3, 1)}
        in_network = {}
        per_kernel = {}
        kv_dim = cfg.dim * cfg.n_kv_heads // cfg.n_heads
        for cls, nm, kid in ((1, "qkv_rmsnorm_rope_q4", 3), (6, "attention+oproj_accum (one launch, fusion level 3)", None),
                             (16, "gemv_q4_hidden_to_dim_accum", 2), (32, "final_rmsnorm+classifier_f16", 5)):
            if (cls == 16 and pair) or (cls == 1 and three):      # the down projection (and, at level 5, every layer's QKV but the first) is inside the FFN launch
                continue
            p0_ = tr.pos()
            a_, mn_, mx_, n_ = tr.bench_in_network(cls, 4)
            in_network[nm] = round(a_, 3)
            if kid is not None:
                nb_ = kb[kid][1]
            else:   # o-proj QWeight + residual in/out + the K and V rows of positions 0..pos of every kv head + q + output
                avg_pos_ = p0_ + 1.5
                nb_ = kb[4][1] + int((avg_pos_ + 1) * 2 * kv_dim * 2) + 2 * cfg.dim * 2
            if cls == 32:       # final rmsnorm + classifier: one launch where the classifier runs as strips with the norm inside (csrc/gemv_strip_cls.h),
                if n_ >= 8:     # else two launches per token (4 steps timed): price the pair
                    a_ = 2.0 * a_
                nb_ += 3 * cfg.dim * 2
            per_kernel[nm] = {"hip_event_us": round(a_, 3), "bytes": nb_, "GBps": round(nb_ / a_ / 1e3, 1),
                              "frac": round(nb_ / a_ / 1e3 / HBM_PEAK_GBS, 4), "launches": n_, "first_position": p0_}
        in_network[kb[dom_id][0]] = round(net_avg, 3)
        per_kernel[kb[dom_id][0]] = {"hip_event_us": round(net_avg, 3), "graph_us": round(graph_us, 3), "bytes": kb[dom_id][1],
                                     "frac": round(kb[dom_id][1] / kernel_us / 1e3 / HBM_PEAK_GBS, 4), "launches": net_n, "first_position": pos_first}
        if pair:    # what the launch replaced, on this box: the launches of fusion level 3, inside the same eager network
            level_now = L.q4_get_fusion()
            L.q4_set_fusion(3)
            gu_, _, _, ngu_ = tr.bench_in_network(8, 8)
            dn_, _, _, ndn_ = tr.bench_in_network(16, 8)
            qk_ = tr.bench_in_network(1, 8)[0] if three else 0.0
            L.q4_set_fusion(level_now)
            per_kernel[kb[dom_id][0]]["replaces"] = {"gate_up_launch_us": round(gu_, 3), "down_launch_us": round(dn_, 3), "qkv_launch_us": round(qk_, 3) if three else None,
                                                     "sum_us": round(gu_ + dn_ + qk_, 3),
                                                     "frac_of_the_sum": round(kb[dom_id][1] / (gu_ + dn_ + qk_) / 1e3 / HBM_PEAK_GBS, 4),
                                                     "note": "fusion level 3's gate/up launch (csrc/gemv_strip.h), down projection (csrc/gemv_q4.h, K split) and, where the launch holds it, "
                                                             "QKV launch by HIP events in the same network"}
        traffic, traffic_src = None, None
        # PMC passes need rocprofv3 around the process: measured separately (tools/profile_round.sh), committed summaries
        for tname in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and args.model == "7b":
                tj_all = json.load(open(tpath))
                tj = tj_all.get("0", tj_all)
                # measured HBM traffic / algorithmic bytes of the other launches of the token (same PMC passes, per kernel)
                prefixes = {"qkv_rmsnorm_rope_q4": "gemv_q4_kernel<1,", "gemv_q4_hidden_to_dim_accum": ("gemv_q4_kernel<0,", "down_strip_kernel<"),
                            "attention+oproj_accum (one launch, fusion level 3)": ("attention_oproj_kernel<", "attention_oproj16_kernel<"), "final_rmsnorm+classifier_f16": ("gemv_f16_kernel<", "cls_strip_kernel<"),
                            kb[dom_id][0]: ("ffn_pair_kernel<true, false, true",) if three else ("ffn_pair_kernel<",) if pair else ("gemv_q4_kernel<2,", "ffn_strip_kernel<", "ffn_strip_pair_kernel<")}
                if three:
                    prefixes.pop("qkv_rmsnorm_rope_q4", None)
                for nm_, pre_ in prefixes.items():
                    ratios = [e_["traffic_over_algorithmic"] for k_, e_ in tj_all.get("%s_n%d" % (args.model, ntok), {}).items()
                              if isinstance(e_, dict) and k_.replace("q4::", "").startswith(pre_) and "traffic_over_algorithmic" in e_]   # (str.startswith takes a tuple too)
                    if ratios and nm_ in per_kernel:
                        per_kernel[nm_]["traffic_over_algorithmic"] = round(max(ratios), 4)
                        per_kernel[nm_]["traffic_source"] = "profiles/" + tname
                if "traffic_bytes_per_launch" in tj:
                    traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)" % tname
                    break
        roofline = {"bound": "hbm", "kernel": kb[dom_id][0], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": kb[dom_id][1], "avg_launch_us": dom["us"],
                    "timing": "kernel duration = max(live HIP-event average, rocprofv3 average of the committed eager trace)",
                    "hip_event_us": round(net_avg, 3), "hip_event_min_us": round(net_min, 3), "hip_event_launches": net_n,
                    "hip_event_frac": round(kb[dom_id][1] / net_avg / 1e3 / HBM_PEAK_GBS, 4),
                    "hip_event_timing": "HIP events (hipExtLaunchKernelGGL start/stop) on the launch stream, every gate/up "
                                        "launch of 16 eager decode steps from position 64 (untraced; excludes the boundary)",
                    "rocprof_avg_us": rocprof_us, "rocprof_source": rocprof_src,
                    "graph_us_per_launch": round(graph_us, 3),
                    "graph_frac": round(kb[dom_id][1] / graph_us / 1e3 / HBM_PEAK_GBS, 4),
                    "graph_timing": "per launch inside a hipGraph (the mode the token loop runs in): 20 replays of a 32-launch "
                                    "graph over the ring of the layers' weights, kernel + boundary",
                    "isolated_ring_us": kernels[kb[dom_id][0]]["us"],
                    "per_kernel": per_kernel}
        # what else bounds the kernel: the committed SQ counters of the same launch (tools/profile_sq.sh), priced at the shader clock read
        # from rocm-smi while the token loop runs. The int4 dequant-dot keeps the VALU pipes busy for 0.43-0.48 of the launch: not the bound
        for tag in ("r06", "r05", "r04"):
            if "valu" in roofline:
                break
            sq_path = os.path.join(ROOT, "profiles", "%s_sq_counters_%s.json" % (tag, "ffn_pair" if pair else "gate_up"))
            if os.path.exists(sq_path) and args.model == "7b":
                sq = json.load(open(sq_path))
                clock_ghz, clock_src = sample_sclk_ghz(lambda: tr.generate_ids(PROMPT_IDS, ntok), device=local_rank)   # shader clock under this load
                if clock_ghz is None:
                    roofline["valu"] = {"note": "no shader clock reading (%s): the committed SQ counters are not priced" % clock_src}
                    break
                busy_us = sq.get("valu_busy_cycles_per_simd", 0.0) / (clock_ghz * 1e3)
                roofline["valu"] = {"instructions_per_launch": sq.get("SQ_INSTS_VALU"), "instructions_per_wave": sq.get("valu_instructions_per_wave"),
                                    "cycles_per_instruction": sq.get("cycles_per_valu_instruction"),
                                    "busy_cycles_per_simd": sq.get("valu_busy_cycles_per_simd"), "sclk_GHz": round(clock_ghz, 3), "sclk_source": clock_src,
                                    "busy_us": round(busy_us, 2),
                                    "busy_fraction_of_launch": round(busy_us / dom["us"], 3),
                                    "reading": ("the VALU pipes are busy for %.2f of the launch" % (busy_us / dom["us"])) + (": below one half the launch is bounded by first-data "
                                               "latency, the x chain, the hand-off and the tail of its stream rather than by VALU issue (DESIGN.md)" if busy_us / dom["us"] < 0.5 else ""),
                                    "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ...)" % os.path.basename(sq_path)}
        kernels["in_network_us"] = in_network
        ids4 = (11, 4) if three else (10, 3, 4) if pair else (0, 2, 3, 4)
        int4_bytes = sum(kb[k][1] for k in ids4)
        int4_us = sum(kernels[kb[k][0]]["us"] for k in ids4)
        kernels["int4_gemv_all_per_layer"] = {"us": round(int4_us, 3), "bytes": int4_bytes, "GBps": round(int4_bytes / int4_us / 1e3, 1)}

    # ---- CPU baseline (rank 0, N == 1 only): the oracle restating run_llama_network ------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle
        m = oracle.Model(path)
        nthreads = oracle.lib().orc_num_threads()
        t0 = time.perf_counter()
        n_done = 0
        ctoks = list(PROMPT_IDS)
        worst = 0.0
        agree = 0
        tr.reset(PROMPT_IDS)
        refs = []
        for pos in range(min(args.cpu_steps, ntok)):
            ref = m.forward(ctoks[pos], pos)
            refs.append(ref)
            n_done += 1
            if pos >= len(PROMPT_IDS) - 1:
                ctoks.append(int(np.argmax(ref.astype(np.float32))))
            if time.perf_counter() - t0 > 40:
                break
        cpu_secs = time.perf_counter() - t0
        # parity of the same positions on the GPU (not timed)
        tr.reset(PROMPT_IDS)
        m2_tokens = []
        for pos in range(n_done):
            tr.run_transformer(pos >= len(PROMPT_IDS) - 1)
            api.synchronize()
        glog = tr.logits().astype(np.float32)
        rlog = ref.astype(np.float32)
        worst = float(np.max(np.abs(glog - rlog) / np.maximum(1.0, np.abs(rlog))))
        agree = int(sum(int(tr.token(i)) == ctoks[i] for i in range(len(PROMPT_IDS), min(len(ctoks), n_done + 1))))
        parity = {"positions": n_done, "last_logits_max_rel_err": round(worst, 5),
                  "greedy_tokens_agree": "%d/%d" % (agree, max(0, min(len(ctoks), n_done + 1) - len(PROMPT_IDS)))}
        # yardstick: the same network in double without intermediate rounding (oracle orc_forward_f64) for the first
        # positions -- how far each of the two fp16 evaluations (GPU, reference-order restatement) is from it
        n64 = min(n_done, args.f64_steps)
        if n64 > 0:
            t64 = time.perf_counter()
            for pos in range(n64):
                exact = m.forward_f64(ctoks[pos], pos, cap=n64)
            tr.reset(PROMPT_IDS)
            for pos in range(n64):
                tr.run_transformer(pos >= len(PROMPT_IDS) - 1)
                api.synchronize()
            g64 = tr.logits().astype(np.float64)
            r64 = refs[n64 - 1].astype(np.float64)
            den = np.maximum(1.0, np.abs(exact))
            parity["vs_unrounded_f64"] = {"position": n64 - 1, "gpu_max_rel_err": round(float(np.max(np.abs(g64 - exact) / den)), 5),
                                          "cpu_restatement_max_rel_err": round(float(np.max(np.abs(r64 - exact) / den)), 5),
                                          "gpu_rms_err": round(float(np.sqrt(np.mean((g64 - exact) ** 2))), 5),
                                          "cpu_restatement_rms_err": round(float(np.sqrt(np.mean((r64 - exact) ** 2))), 5),
                                          "argmax_equal": bool(int(np.argmax(g64)) == int(np.argmax(exact))),
                                          "seconds": round(time.perf_counter() - t64, 1)}
        # config 1 beside it: the restated single int4 GEMV 4096 -> 11008 on the host cores (median of 20 calls)
        from llama_cu_awq_amd import synth as _synth
        rng = np.random.default_rng(20240229)
        gw, gz, gs = _synth.random_qweight(rng, cfg.dim, cfg.hidden_dim)
        gx = rng.standard_normal(cfg.dim).astype(np.float16)
        gt = []
        for _ in range(20):
            tg = time.perf_counter()
            oracle.matmul_q4(gx, gw, gz, gs, cfg.dim, cfg.hidden_dim)
            gt.append(time.perf_counter() - tg)
        gemv_cpu_us = 1e6 * float(np.median(gt))
        cpu = {"value": round(n_done / cpu_secs, 4), "unit": "tokens/s", "cores": nthreads, "kind": "port",
               "single_gemv": {"shape": "%d->%d" % (cfg.dim, cfg.hidden_dim), "us": round(gemv_cpu_us, 1),
                               "GBps": round(kb[1][1] / gemv_cpu_us / 1e3, 2)},
               "sample": "%d decode steps (positions 0..%d) of the same %s checkpoint, CPU restatement of run_llama_network "
                         "(oracle/q4_oracle.c, OpenMP); the reference has no CPU path" % (n_done, n_done - 1, args.model)}
        m.close()

    # ---- BASELINE configs 3 and 4 beside the headline (rank 0, N == 1): 13B -n 256 and 7B -n 2048, same rule ------------
    extra = None
    GB_PER_TOKEN = {("7b", 256): 3.695, ("13b", 256): 7.026, ("7b", 2048): 4.164}      # weights + average KV traffic
    if rank == 0 and world == 1 and not args.no_extra and args.model == "7b" and ntok == 256:
        extra = {}
        tr.close()
        for mname, n in (("7b", 2048), ("13b", 256), ("13b", 2048)):
            g2 = synth.GEOMETRIES[mname]
            p2 = os.path.join(args.model_dir, "llama2_q4_synth_%s_seed20240229.bin" % mname)
            if not (os.path.exists(p2) and os.path.getsize(p2) == synth.model_bytes(g2)):
                synth.write_model(p2, g2)
            t2 = api.Transformer(p2, temperature=0.0)
            t2.generate_ids(PROMPT_IDS, n)                       # warm + graph capture of every bin
            runs = [t2.generate_ids(PROMPT_IDS, n) for _ in range(2)]
            best = max(r[1] for r in runs)
            kv_dim2 = g2[0] * g2[4] // g2[3]
            streamed2 = synth.model_bytes(g2) - 32 - g2[5] * g2[0] * 2            # every tensor but the embedding table (one row of it per token)
            gb = GB_PER_TOKEN.get((mname, n), round((streamed2 + 2 * g2[2] * (n / 2.0 + 0.5) * kv_dim2 * 2) / 1e9, 3))
            extra["llama2_%s_n%d" % (mname, n)] = {"tokens_per_s": round(best, 1), "ms_per_token": round(1000.0 / best, 4), "GB_per_token": gb,
                                                    "frac_of_8TBps": round(best * gb / HBM_PEAK_GBS, 4)}
            if n == 2048:
                # the KV-stress regime alone: positions 1024..2047 (the last sequence-length bin), from the difference of two generations'
                # wall times; bytes per token there = weights + K and V rows of 1536.5 positions on average
                half = min(t2.generate_ids(PROMPT_IDS, 1024)[3] for _ in range(2))
                full = min(r[3] for r in runs)
                tps_hi = 1024.0 / (full - half)
                gb_hi = (streamed2 + 2 * g2[2] * 1536.5 * kv_dim2 * 2) / 1e9
                extra["llama2_%s_n%d" % (mname, n)]["positions_1024_2047"] = {"tokens_per_s": round(tps_hi, 1), "ms_per_token": round(1000.0 / tps_hi, 4),
                                                                              "GB_per_token": round(gb_hi, 3), "frac_of_8TBps": round(tps_hi * gb_hi / HBM_PEAK_GBS, 4)}
            if mname == "13b" and n == 256:
                # the 13B launches by HIP events inside the eager network, like roofline.per_kernel above (gate/up and the down projection run as
                # strips there: csrc/gemv_strip.h, gemv_strip_down.h)
                class _C2:
                    pass
                c2 = _C2()
                c2.dim, c2.hidden_dim, c2.vocab_size = g2[0], g2[1], g2[5]
                kb2 = kernel_bytes(c2)
                t2.reset(PROMPT_IDS)
                for pos in range(64):
                    t2.run_transformer(pos >= len(PROMPT_IDS) - 1)
                    api.synchronize()
                pk = {}
                for cls, nm, kid in ((8, kb2[0][0], 0), (1, "qkv_rmsnorm_rope_q4", 3), (16, "gemv_q4_hidden_to_dim_accum", 2)):
                    a_, mn_, mx_, n_ = t2.bench_in_network(cls, 8)
                    pk[nm] = {"hip_event_us": round(a_, 3), "bytes": kb2[kid][1], "GBps": round(kb2[kid][1] / a_ / 1e3, 1),
                              "frac": round(kb2[kid][1] / a_ / 1e3 / HBM_PEAK_GBS, 4), "launches": n_}
                extra["llama2_%s_n%d" % (mname, n)]["per_kernel"] = pk
            t2.close()
        # the reference CLI's DEFAULT mode (-t 0.5 -p 0.6, llama2_q4.cu:632-633: temperature / top-p sampling inside the captured graphs,
        # csrc/q4_sampling.hip) on the headline geometry, and the grouped-query geometry of Mistral-7B (llama2_q4.cu:309-313), greedy
        t3 = api.Transformer(path, temperature=0.5, topp=0.6, seed=20240229)
        t3.generate_ids(PROMPT_IDS, ntok)
        r3 = [t3.generate_ids(PROMPT_IDS, ntok) for _ in range(2)]
        b3 = max(r3, key=lambda r: r[1])
        extra["llama2_7b_n256_t0.5_p0.6"] = {"tokens_per_s": round(b3[1], 1), "ms_per_token": round(1000.0 / b3[1], 4), "timed_tokens": int(b3[2]),
                                              "GB_per_token": GB_PER_TOKEN[("7b", 256)], "frac_of_8TBps": round(b3[1] * GB_PER_TOKEN[("7b", 256)] / HBM_PEAK_GBS, 4),
                                              "sampler": "temperature 0.5, top-p 0.6, seed 20240229 (a sampled EOS ends a generation early: timed_tokens)"}
        t3.close()
        g4 = synth.GEOMETRIES["mistral7b"]
        p4 = os.path.join(args.model_dir, "llama2_q4_synth_mistral7b_seed20240229.bin")
        if not (os.path.exists(p4) and os.path.getsize(p4) == synth.model_bytes(g4)):
            synth.write_model(p4, g4)
        t4 = api.Transformer(p4, temperature=0.0)
        t4.generate_ids(PROMPT_IDS, 256)
        b4 = max(t4.generate_ids(PROMPT_IDS, 256)[1] for _ in range(2))
        kv4 = g4[0] * g4[4] // g4[3]
        gb4 = (synth.model_bytes(g4) - 32 - g4[5] * g4[0] * 2 + 2 * g4[2] * 128.5 * kv4 * 2) / 1e9     # weights without the embedding table + average K / V rows
        extra["mistral7b_n256"] = {"tokens_per_s": round(b4, 1), "ms_per_token": round(1000.0 / b4, 4), "GB_per_token": round(gb4, 3),
                                   "frac_of_8TBps": round(b4 * gb4 / HBM_PEAK_GBS, 4), "geometry": "dim 4096, hidden 14336, 32 layers, 32 heads / 8 kv heads"}
        t4.close()
        tr = api.Transformer(path, temperature=0.0)

    if rank == 0:
        name, cus, mem = api.device_info()
        out = {
            "metric": "decode_tokens_per_sec", "value": round(value, 2), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / max(args.steps, 1), 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / PUBLISHED_TOKS, 3),
            "baseline_note": "published 200.79 tok/s is the reference on an RTX 4090 (README.md:111); no MI355X number exists",
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "Llama-2-%s AWQ w4 g128 (synthetic weights, real geometry), -n %d greedy, 8-token prompt, "
                                   "%d timed tokens per step" % (args.model.upper(), ntok, timed_tokens // max(args.steps, 1)),
                       "parallelism": "replicas x%d (no data-path collective)" % world},
            "ms_per_token": round(1000.0 * elapsed * world / max(total_tokens, 1), 4),
            "whole_token": {"GB_per_token": GB_PER_TOKEN.get((args.model, ntok)), "frac_of_8TBps": round(value / max(world, 1) * GB_PER_TOKEN[(args.model, ntok)] / HBM_PEAK_GBS, 4)
                            if (args.model, ntok) in GB_PER_TOKEN else None},
            "roofline": roofline, "cpu_baseline": cpu, "extra_configs": extra, "kernels": kernels, "parity_vs_cpu_restatement": parity,
            "device": name, "cus": cus, "load_s": round(t_load, 2), "synth_s": round(t_gen, 2),
            "tok_s_per_generation": [round(x, 1) for x in per_gen],
        }
        print(json.dumps(out))
    tr.close()
    L.q4_set_stream(None)
    L.q4_stream_destroy(stream)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
