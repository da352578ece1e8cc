"""End-to-end parity: build_transformer + run_transformer (graphs, fused and 1:1 kernel sequences) against the CPU
restatement of run_llama_network on synthetic checkpoints (tiny / GQA / small), through the C ABI."""
import os

import numpy as np
import pytest

from conftest import f16_ulp_diff
from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    d = tmp_path_factory.mktemp("models")
    out = {}
    for name in ("tiny", "tiny_gqa", "small", "longk_gqa", "head128", "head128_k5120", "head64_long", "head128_gqa", "tinyllama", "head256",
                 "head128_k8192", "head96", "head80_gqa"):
        p = str(d / (name + ".bin"))
        synth.write_model(p, name, seed=7)
        out[name] = p
    return out


# measured worst case per model (tools/measure_tolerances.py, profiles/r02_parity_observed.json): one fp16 ulp of an O(1)
# logit = 9.8e-4 (longk_gqa, logits up to 2.2: 1.7e-3); the bounds are 3x that
BOUND = {"tiny": 3e-3, "tiny_gqa": 3e-3, "small": 3e-3, "longk_gqa": 5e-3, "head128": 5e-3, "head128_k5120": 5e-3, "head128_gqa": 5e-3,
         "tinyllama": 5e-3, "head256": 5e-3, "head128_k8192": 1.5e-2, "head96": 3e-3, "head80_gqa": 3e-3}   # K = 8192: measured 5.0e-3 (5 fp16 ulps of a logit below 1)


def _logit_close(gpu, ref, bound=5e-3):
    gpu, ref = gpu.astype(np.float64), ref.astype(np.float64)
    return np.abs(gpu - ref) <= bound * np.maximum(1.0, np.abs(ref))


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa", "small", "longk_gqa", "head128", "head128_k5120", "head128_gqa", "tinyllama", "head256",
                                  "head128_k8192", "head96", "head80_gqa"])
@pytest.mark.parametrize("fusion,graphs", [(5, 1), (3, 1), (1, 1), (0, 1), (3, 0), (1, 0), (0, 0)])
def test_forward_logits_and_kv(q4, orc, models, name, fusion, graphs):
    L = q4.lib()
    L.q4_set_fusion(fusion)
    L.q4_set_use_graphs(graphs)
    try:
        t = q4.Transformer(models[name])
        m = orc.Model(models[name])
        prompt = [1, 17, 300, 45, 9]
        steps = 12
        t.reset(prompt)
        toks = list(prompt)
        for pos in range(steps):
            gen = pos >= len(prompt) - 1
            t.run_transformer(gen)
            q4.synchronize()
            ref = m.forward(toks[pos], pos)
            got = t.logits()
            assert _logit_close(got, ref, BOUND[name]).all(), "pos %d: max |d| %g" % (pos, np.abs(got.astype(np.float32) - ref.astype(np.float32)).max())
            assert t.pos() == pos + 1
            if gen:
                nxt = t.token(pos + 1)
                top2 = np.sort(ref.astype(np.float32))[-2:]
                if top2[1] - top2[0] > 4e-3 * max(1.0, abs(top2[1])):      # not a near-tie
                    assert nxt == int(np.argmax(ref.astype(np.float32))), pos
                toks.append(nxt)
        rk, rv = m.kv()
        for layer in range(t.config.n_layers):
            for pos in (0, steps - 1):
                gk, gv = t.kv_row(layer, pos)
                assert _logit_close(gk, rk[layer, pos], 3e-3).all() and _logit_close(gv, rv[layer, pos], 3e-3).all(), (layer, pos)
        t.close()
        m.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
        L.q4_set_use_graphs(1)


@pytest.mark.parametrize("name,steps,checkpoints", [("head128", 1060, (3, 100, 127, 128, 200, 255, 256, 300, 511, 512, 600, 1023, 1024, 1059)),
                                                     ("head128_k5120", 290, (3, 127, 128, 200, 255, 256, 289)),
                                                     ("head128_gqa", 690, (3, 100, 127, 128, 255, 256, 511, 512, 689)),
                                                     ("tinyllama", 1060, (3, 100, 127, 128, 200, 255, 256, 300, 511, 512, 600, 1023, 1024, 1059)),
                                                     ("head256", 590, (3, 100, 127, 128, 255, 256, 511, 512, 589)),
                                                     ("head128_k8192", 290, (3, 127, 128, 200, 255, 256, 289))])
def test_in_launch_handoffs_reproduce_the_launch_sequence_bits(q4, models, observed, name, steps, checkpoints):
    """Fusion level 3 (attention -> o-proj as ONE launch, the hand-off inside the launch) against levels 1 and 0 across the
    sequence-length bins 128 / 256 (one block per head) and 512 / 1024 / seq_len (split context), for heads of 64 / 128 / 256,
    multi-head and grouped-query, K = dim in one, two, three (shared half slot) and four k-slots: the same arithmetic with the
    same rounding points, so the logits must agree within the model's bound with equal greedy token rings (until a near-tie),
    levels 1 and 0 bit for bit, and no bounded spin may have run out."""
    L = q4.lib()
    outs = {}
    timeouts_before = L.q4_handoff_timeouts()     # process-wide counter: compare against a snapshot
    try:
        for fusion in (3, 1, 0):
            L.q4_set_fusion(fusion)
            t = q4.Transformer(models[name])
            t.reset([1, 5, 9])
            got = []
            for pos in range(steps):
                t.run_transformer(pos >= 2)
                if pos in checkpoints:
                    q4.synchronize()
                    got.append(t.logits().view(np.uint16).copy())
            q4.check(L.q4_handoff_status(t.state))
            assert L.q4_get_fusion() == fusion and L.q4_handoff_timeouts() == timeouts_before
            ring = [int(t.token(i)) for i in range(steps + 1)]
            outs[fusion] = (got, ring)
            t.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    # levels 1 and 3 run the same device code with another fp32 grouping of an output's positions: below bin 512 the fused
    # launch's attention role works with head_size / 32 blocks per head, 16 positions per wave instruction of a block's V slice
    # (the stand-alone kernel: 4), and with 8 waves where the stand-alone kernel of the other head sizes has 16 -- so the
    # comparison is the model's bound with equal greedy token rings. (The one-block form of the role, which reproduces level 1
    # bit for bit in the first bin, is compared in tests/prof_cases.py.)
    first_div = next((i for i, (x, y) in enumerate(zip(outs[1][1], outs[3][1])) if x != y), None)
    observed.setdefault("fusion3_vs_1_first_token_divergence", {})[name] = first_div
    # a near-tie may resolve the other way under another fp32 grouping, but not inside the first bin: measured first divergences
    # 300+ or none (profiles/r04_parity_observed.json)
    assert first_div is None or first_div >= 128, "token rings diverged at %d (fusion 3 vs 1)" % first_div
    for a, b, pos in zip(outs[1][0], outs[3][0], checkpoints):
        if first_div is not None and pos >= first_div:
            break
        af, bf = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
        err = float((np.abs(af - bf) / np.maximum(1.0, np.abs(bf))).max())
        assert err <= (5e-3 if pos <= 128 else 1.2e-2), (pos, err)   # (the two histories differ by an ulp from early on: measured 5.4e-3 at 256)
    # level 0 (the reference's 1:1 sequence): identical too, except where K = dim ends in a shared half slot (K = 5120):
    # there a column's half-slot terms sit in the lower or the upper 32 lanes depending on its place in the wave, and the
    # RoPE-paired column order of the fused QKV differs from the plain one -- same terms, another fp32 rounding sequence
    for a, b, pos in zip(outs[0][0], outs[1][0], checkpoints):
        if name == "head128_k5120":
            # a single fp16 value of the residual stream rounding the other way (one ulp of a unit-scale value = 9.8e-4 .. 2e-3) moves every
            # logit by up to that much: bounded in units of a unit-scale ulp, not of the logit's own (a logit of 1e-5 would read as 1000 ulps)
            af, bf = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
            big = np.abs(bf) >= 1.0
            # logits of unit scale and above: in ulps of the logit itself (the old exact check's successor must still see a regression there);
            # below: absolute. Bounds = twice the measured maxima (1 ulp, 9.8e-4: profiles/r06_parity_observed.json, "fusion0_vs_1_head128_k5120")
            ulp = np.spacing(np.abs(bf).astype(np.float16)).astype(np.float64)
            ulps_big = float((np.abs(af - bf)[big] / ulp[big]).max()) if big.any() else 0.0
            abs_small = float(np.abs(af - bf)[~big].max()) if (~big).any() else 0.0
            observed.setdefault("fusion0_vs_1_head128_k5120", {})[str(pos)] = {"max_ulps_of_logits_above_1": ulps_big, "max_abs_below_1": abs_small,
                                                                                "share_of_logits_that_differ": float((af != bf).mean())}
            assert ulps_big <= 2.0 and abs_small <= 2e-3 and (af != bf).mean() <= 0.6, (pos, ulps_big, abs_small, float((af != bf).mean()))
            break                                       # later positions follow their own greedy tokens
        assert np.array_equal(a, b), "logits differ at position %d (fusion 0 vs 1)" % pos
    if name != "head128_k5120":
        assert outs[0][1] == outs[1][1], "token ring differs at fusion level 0"


@pytest.mark.parametrize("name", ["head128", "head64_long"])
def test_a_second_sequence_never_meets_the_first_ones_records(q4, models, name):
    """Two DIFFERENT prompts of different lengths back to back on one Transformer through the split-context bins (>= 512), at
    fusion level 3: the tagged words that outlive a launch (attention granules, flash-decode records in RunState::att) validate
    themselves by tag alone, so the second sequence must never re-create a tag of the first (the epoch word is not rewound
    between sequences). Sequence B after sequence A must equal sequence B on a fresh model, bit for bit, and stay within the
    model's bound of fusion level 1."""
    L = q4.lib()
    before = L.q4_handoff_timeouts()
    prompt_a, steps_a = [1, 5, 9], 700
    prompt_b, steps_b = [1, 44, 7, 300, 12, 90, 3], 760
    checkpoints = (520, 640, 698, 699, 700, 759)

    def run(t, prompt, steps, marks=()):
        t.reset(prompt)
        got = []
        for pos in range(steps):
            t.run_transformer(pos >= len(prompt) - 1)
            if pos in marks:
                q4.synchronize()
                got.append(t.logits().view(np.uint16).copy())
        q4.check(L.q4_handoff_status(t.state))
        return got, [int(t.token(i)) for i in range(steps + 1)]

    try:
        L.q4_set_fusion(3)
        t = q4.Transformer(models[name])
        run(t, prompt_a, steps_a)
        after_a = run(t, prompt_b, steps_b, checkpoints)
        t.close()
        t = q4.Transformer(models[name])
        fresh = run(t, prompt_b, steps_b, checkpoints)
        t.close()
        L.q4_set_fusion(1)
        t = q4.Transformer(models[name])
        level1 = run(t, prompt_b, steps_b, checkpoints)
        t.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    assert L.q4_handoff_timeouts() == before
    assert after_a[1] == fresh[1], "the second sequence's tokens depend on the sequence before it"
    for a, b, pos in zip(after_a[0], fresh[0], checkpoints):
        assert np.array_equal(a, b), "logits of the second sequence depend on the sequence before it (position %d)" % pos
    for a, b, pos in zip(after_a[0], level1[0], checkpoints):
        if after_a[1][:pos + 1] != level1[1][:pos + 1]:
            break                                        # a near-tie resolved the other way: later positions follow other tokens
        af, bf = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
        assert float((np.abs(af - bf) / np.maximum(1.0, np.abs(bf))).max()) <= 1.2e-2, pos


@pytest.mark.parametrize("name,target", [("head128", 1050), ("head64_long", 1100), ("head64_long", 1290), ("head128_gqa", 600), ("head128_gqa", 300)])
def test_split_context_merge_by_the_last_block(q4, orc, models, name, target):
    """Bins >= 1024 inside the network: one attention block per (head, 256 positions), merged by each head's LAST block
    (returning arrival on the model's counters, no second launch). Decode `target` positions through the captured graphs,
    hand the GPU's KV cache to the restatement and compare ONE step from the identical state (head 128, and head 64 with
    grouped-query attention)."""
    import ctypes as C
    L = q4.lib()
    t = q4.Transformer(models[name])
    m = orc.Model(models[name])
    toks, tps, timed, _ = t.generate_ids([1, 5, 9], target)
    assert t.pos() == target
    cfg = t.config
    n = cfg.n_layers * cfg.seq_len * (cfg.dim * cfg.n_kv_heads // cfg.n_heads)
    k = np.empty(n, dtype=np.uint16)
    v = np.empty(n, dtype=np.uint16)
    q4.check(L.q4_memcpy_d2h(k.ctypes.data, t.state.contents.key_cache, k.nbytes))
    q4.check(L.q4_memcpy_d2h(v.ctypes.data, t.state.contents.value_cache, v.nbytes))
    np.ctypeslib.as_array(m.L.orc_key_cache(m.h), shape=(n,))[:] = k
    np.ctypeslib.as_array(m.L.orc_value_cache(m.h), shape=(n,))[:] = v
    tok = int(t.token(target))
    for rep in range(3):                                   # three consecutive steps: the counters re-arm themselves
        t.run_transformer(True)
        q4.synchronize()
        got = t.logits()
        ref = m.forward(tok, target + rep)
        assert _logit_close(got, ref, 5e-3).all(), (rep, np.abs(got.astype(np.float32) - ref.astype(np.float32)).max())
        tok = int(t.token(target + rep + 1))
        top2 = np.sort(ref.astype(np.float32))[-2:]
        if top2[1] - top2[0] > 4e-3 * max(1.0, abs(top2[1])):
            assert tok == int(np.argmax(ref.astype(np.float32)))
        else:
            break
    t.close()
    m.close()


@pytest.mark.parametrize("name", ["small", "tiny_gqa", "longk_gqa"])
def test_fused_equals_unfused_bits(q4, models, name):
    """The fused sequence must reproduce the 1:1 kernel chain (same canonical reductions): identical logits -- also
    for grouped-query attention, where the reference runs three GEMVs (llama2_q4.cu:310-312) and the fused QKV launch
    sizes its k/v part to kv_dim."""
    L = q4.lib()
    outs = []
    for fusion in (0, 1):
        L.q4_set_fusion(fusion)
        t = q4.Transformer(models[name])
        t.reset([1, 5, 9])
        for pos in range(6):
            t.run_transformer(pos >= 2)
        q4.synchronize()
        outs.append(t.logits().copy())
        t.close()
    L.q4_set_fusion(q4.DEFAULT_FUSION)
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))


def test_generate_ids_matches_oracle(q4, orc, models):
    t = q4.Transformer(models["small"])
    m = orc.Model(models["small"])
    prompt = [1, 400, 22, 7, 513, 99, 1000, 3]
    steps = 48
    gtoks, tps, timed, secs = t.generate_ids(prompt, steps)
    rtoks, rlogits = m.generate_greedy(prompt, steps, want_logits=True)
    assert timed == steps - 1 and tps > 0
    # identical until the first near-tie in the oracle's logits
    n = min(len(gtoks), len(rtoks))
    for i in range(n):
        if gtoks[i] != rtoks[i]:
            lg = np.sort(rlogits[i - 1])[-2:]
            assert lg[1] - lg[0] < 4e-3 * max(1.0, abs(lg[1])), "token %d differs without a near-tie" % i
            break
    t.close()
    m.close()


def test_perplexity_path(q4, orc, models):
    t = q4.Transformer(models["small"], perplexity=True)
    m = orc.Model(models["small"])
    rng = np.random.default_rng(3)
    toks = np.concatenate([[1], rng.integers(3, 1024, size=40)]).astype(np.int32)
    ppl = t.perplexity_ids(toks)
    glog = t.logits_array(40)
    rlog = np.stack([m.forward(int(toks[i]), i).astype(np.float32) for i in range(40)])
    assert (np.abs(glog - rlog) <= 3e-3 * np.maximum(1.0, np.abs(rlog))).all()
    rppl = orc.compute_perplexity(toks[1:41], rlog)
    assert abs(ppl - rppl) <= 5e-3 * rppl, (ppl, rppl)      # perplexity within 0.5 %
    t.close()
    m.close()


def test_pipelined_generate_equals_stepwise_run_transformer(q4, models):
    """generate() queues step pos before it has seen step pos-1's token (q4_run_transformer_at + q4_wait_pos); the
    token ring must equal the reference order: synchronise, run_transformer, read (llama2_q4.cu:468-474)."""
    prompt = [1, 400, 22, 7, 513]
    steps = 40
    t = q4.Transformer(models["small"])
    gtoks, tps, timed, secs = t.generate_ids(prompt, steps)
    t.reset(prompt)
    ring = list(prompt)
    for pos in range(steps):
        q4.synchronize()
        t.run_transformer(pos >= len(prompt) - 1)
        q4.synchronize()
        if pos + 1 >= len(prompt):
            ring.append(int(t.token(pos + 1)))
    n = min(len(gtoks), len(ring))
    assert n >= steps - 1
    stop = next((i for i in range(1, n) if ring[i] == 2), n)       # generate() stops at EOS
    assert list(gtoks[:stop]) == ring[:stop]
    t.close()


def test_bench_in_network_times_the_products_own_launches(q4, models):
    t = q4.Transformer(models["small"])
    prompt = [1, 400, 22, 7]
    t.reset(prompt)
    for pos in range(6):
        t.run_transformer(pos >= len(prompt) - 1)
        q4.synchronize()
    before = int(t.pos())
    avg, mn, mx, n = t.bench_in_network(8, tokens=3)
    assert n == 3 * t.config.n_layers and 0 < mn <= avg <= mx < 1e4
    assert int(t.pos()) == before + 3                       # three real decode steps were taken
    avg_all, _, _, n_all = t.bench_in_network(1 | 2 | 4 | 8 | 16, tokens=1)
    assert n_all == 4 * t.config.n_layers and avg_all > 0   # fusion level 3: attention + o-proj are one launch (head 64 too)
    q4.lib().q4_set_fusion(1)
    try:
        assert t.bench_in_network(1 | 2 | 4 | 8 | 16, tokens=1)[3] == 5 * t.config.n_layers
    finally:
        q4.lib().q4_set_fusion(q4.DEFAULT_FUSION)
    t.close()


@pytest.mark.parametrize("name", ["small", "tiny_gqa"])
def test_gpu_is_as_close_to_unrounded_arithmetic_as_the_restatement(q4, orc, models, name):
    """Both fp16 evaluations (HIP path, reference-order restatement) against the double, never-rounded forward of the
    same network: the HIP path's error must be of the restatement's size (<= 2x + 1e-3), position by position."""
    t = q4.Transformer(models[name])
    m = orc.Model(models[name])
    prompt = [1, 400, 22, 7, 513, 99]
    t.reset(prompt)
    for pos, tok in enumerate(prompt):
        t.run_transformer(False)
        q4.synchronize()
        g = t.logits().astype(np.float64)
        r = m.forward(tok, pos).astype(np.float64)
        e = m.forward_f64(tok, pos, cap=8)
        den = np.maximum(1.0, np.abs(e))
        eg, er = np.max(np.abs(g - e) / den), np.max(np.abs(r - e) / den)
        assert eg <= 2.0 * er + 1e-3, (pos, eg, er)
    t.close()
    m.close()


def test_sampler_rng_after_an_early_stop_matches_one_draw_per_executed_step(q4, models):
    """sample() draws one coin per run_transformer call (sampler.h:45), also on greedy steps. The token loops queue steps eight
    at a time; when EOS ends the generation inside such a group (the loop looks at tokens[pos], llama2_q4.cu:473-477 -- a prompt
    that contains token 2 stops there) the surplus steps' draws are taken back, so a Sampler that is used again continues
    exactly like the reference's: pos + 1 draws after a stop at position pos."""
    import ctypes as C

    class SamplerStruct(C.Structure):
        _fields_ = [("vocab_size", C.c_int), ("indices", C.c_void_p), ("tempStorage_scan", C.c_void_p), ("tempStorage_sort", C.c_void_p),
                    ("temp_storage_bytes_scan", C.c_size_t), ("temp_storage_bytes_sort", C.c_size_t), ("temperature", C.c_float),
                    ("topp", C.c_float), ("rng_state", C.c_ulonglong)]

    L = q4.lib()
    seed = 1234
    for stop, temperature in ((5, 0.0), (1, 0.0), (7, 0.0), (10, 0.0), (5, 0.7), (12, 0.7)):   # inside the first group, at its edge, inside the second; sampled groups too
        prompt = [1] + [5 + i for i in range(15)]
        prompt[stop] = 2
        t = q4.Transformer(models["tiny"], seed=seed, temperature=temperature, topp=0.9)
        toks, tps, timed, secs = t.generate_ids(prompt, 40)
        assert timed + 1 == stop                                     # pos at the break (timed_tokens = pos - 1, :488)
        state = C.c_ulonglong(seed)
        for _ in range(stop + 1):
            L.random_u32(C.byref(state))
        got = C.cast(t.sampler, C.POINTER(SamplerStruct)).contents.rng_state
        assert got == state.value, (stop, got, state.value)
        t.close()


def test_two_models_alternate_without_recapture(q4, models):
    """Captured graphs are kept per model (csrc/q4_runtime.hip, GraphSet): a host that alternates two Transformers step by step replays each one's
    graphs instead of capturing them again on every switch (q4_graph_captures counts), and every token equals the model's own stepwise run. A fifth
    live model evicts the least recently used set, which is captured again on its next turn."""
    L = q4.lib()
    names = ["tiny", "tiny_gqa"]
    prompt = [1, 17, 300, 45, 9]
    want = {}
    for n in names:                                   # each model alone
        t = q4.Transformer(models[n])
        want[n] = list(t.generate_ids(prompt, 34)[0])
        t.close()
    ts = {n: q4.Transformer(models[n]) for n in names}
    for n in names:
        ts[n].reset(prompt)
    c0 = L.q4_graph_captures()
    for pos in range(30):
        for n in names:                               # 30 alternating steps
            ts[n].run_transformer(pos >= len(prompt) - 1)
    q4.synchronize()
    used = L.q4_graph_captures() - c0
    # tiny models: one bin (seq_len 64), two variants each (prompt steps, generated steps) -> four captures for sixty alternating steps
    assert used == 4, used
    for n in names:
        assert [int(ts[n].token(i)) for i in range(31)] == want[n][:31], n
    # more live models than the cache holds: still correct, the evicted model captures again
    extra = [q4.Transformer(models[x]) for x in ("small", "head96", "head80_gqa")]
    for t in extra:
        t.reset(prompt)
        t.run_transformer(False)
    q4.synchronize()
    c1 = L.q4_graph_captures()
    ts["tiny"].run_transformer(True)                  # `tiny` was the least recently used of five: its set is gone
    q4.synchronize()
    assert L.q4_graph_captures() == c1 + 1
    assert int(ts["tiny"].token(31)) == want["tiny"][31]
    for t in list(ts.values()) + extra:
        t.close()
