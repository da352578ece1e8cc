"""BASELINE.json configs 2-5 at their REAL geometries (synthetic weights, seed 20240229), HIP path through the C ABI
against the CPU restatement (oracle/q4_oracle.c) and the unrounded double forward:

  config 2  Llama-2-7B  -n 256 greedy ........ 32 positions, logits + 25 greedy tokens (run_llama_network llama2_q4.cu:286-340),
            every position against the unrounded double forward as well
  config 3  Llama-2-13B -n 256 greedy ........ 24 positions (17 greedy tokens), the first 12 against the double forward
  config 4  Llama-2-7B  seq_len 2048 ......... captured graphs of every sequence-length bin (llama2_q4.cu:356-360): the GPU's
            own KV cache is copied into the restatement and ONE step is compared from the identical state at positions
            400 (bin 512), 900 (bin 1024), 1100 (bin 2048) and 2040 (end of the context) -- split-context attention throughout
  config 5  Llama-2-7B perplexity path ....... 64 teacher-forced positions, fp32 logits + perplexity (perplexity.h:57-97)
  (+ a Mistral-7B-shaped grouped-query model, 24 positions / 17 greedy tokens: llama2_q4.cu:309-313)
  (+ 50 repeated -n 256 generations of the default path: identical token rings, no timed-out in-launch wait)

Tolerances. At 32-40 layers of RANDOM weights two valid fp16 evaluations of the network drift apart by a few 1e-2 of
max(1,|logit|); the yardstick is the same network in double without rounding (orc_forward_f64): the HIP path may be at
most 2x as far from it as the reference-order restatement (+1e-3). Against the restatement itself the bound is 2x the
measured worst case PER GEOMETRY (BOUND below; profiles/r04_parity_observed.json keeps every position's value). Tokens must be
equal except at near-ties of the restatement's top two logits."""
import ctypes as C
import os

import numpy as np
import pytest

from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu

PROMPT = [1, 2436, 385, 3686, 388, 1048, 22796, 118]      # "write an essay about GPUs", reference tokenizer
PROMPT_SMALL = [1, 17, 300, 45, 9]                         # for the 512-entry vocabularies of the test-size models
MODEL_DIR = os.environ.get("Q4_MODEL_DIR", "/tmp")
# logits vs the reference-order restatement, max |d| / max(1, |logit|), per geometry: 2x the worst value the round's runs measured
# (profiles/r04_parity_observed.json: 7B 0.050 over 32 positions and 0.033 at the config-4 checkpoints, 13B 0.068 over 24 positions
# and 0.051 at position 200, Mistral-shaped 0.041, perplexity path 0.046)
# NOTE: these exceed SURVEY 8(c)'s adopted 3e-2. That figure was set before anything was measured; on 32 (40) layers of random-weight fp16
# arithmetic two VALID evaluations of the reference's own kernels -- the restatement in the reference's lane order and the same sums in another
# fp32 order -- already sit 0.03-0.05 from the never-rounded forward and up to 0.05 from each other (tools/error_growth.py: its control run),
# because every fp16 rounding that falls the other way is amplified by the layers behind it. The principled check is therefore the bracket
# below each comparison: the HIP path may be at most 1.5x (+1e-3) as far from the unrounded double forward as the restatement is (measured
# after round 5's fix of the truncating accumulate: rms ratio 1.02-1.06, the control's own 1.05).
BOUND = {"7b": 0.10, "13b": 0.14, "mistral7b": 0.085}


def _model(name):
    geom = synth.geometry(name)
    path = os.path.join(MODEL_DIR, "llama2_q4_synth_%s_seed20240229.bin" % name)      # the file bench.py uses
    if not (os.path.exists(path) and os.path.getsize(path) == synth.model_bytes(geom)):
        synth.write_model(path, name)
    return path


@pytest.fixture(scope="module")
def m7b():
    return _model("7b")


def _rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _ring(t):
    """The pinned token ring SharedData::tokens (common.h:51-54) as a writable numpy view."""
    base = t.state.contents.shared_data
    return np.ctypeslib.as_array((C.c_int * t.config.seq_len).from_address(base + 4))


def _lockstep(q4, t, m, prompt, steps, f64_steps, bound_vs_restatement, rec=None):
    """Feed both evaluations the same tokens; returns (worst rel err vs restatement, tokens compared, near-tie count)."""
    t.reset(prompt)
    ring = _ring(t)
    toks = list(prompt)
    worst, compared, ties = 0.0, 0, 0
    per_pos = []
    for pos in range(steps):
        gen = pos >= len(prompt) - 1
        t.run_transformer(gen)
        q4.synchronize()
        ref = m.forward(toks[pos], pos)
        got = t.logits()
        e = _rel(got, ref)
        worst = max(worst, e)
        per_pos.append(round(e, 5))
        assert e <= bound_vs_restatement, "pos %d: max rel logit err %g vs the restatement" % (pos, e)
        if pos < f64_steps:
            exact = m.forward_f64(toks[pos], pos, cap=f64_steps)
            eg, er = _rel(got, exact), _rel(ref, exact)
            # the worst of 32000 logits is a noisy statistic (two valid fp32 orders of the reference's own sums differ by as much between
            # themselves, tools/error_growth.py: control run): 2x on the maximum, 1.5x on the rms over the vocabulary
            assert eg <= 2.0 * er + 1e-3, "pos %d: GPU %g vs restatement %g from the unrounded forward" % (pos, eg, er)
            rg, rr = _rms(got, exact), _rms(ref, exact)
            assert rg <= 1.5 * rr + 1e-4, "pos %d: rms distance from the unrounded forward, GPU %g vs restatement %g" % (pos, rg, rr)
            if rec is not None:
                rec["vs_f64_gpu_max"] = max(rec.get("vs_f64_gpu_max", 0.0), eg)
                rec["vs_f64_restatement_max"] = max(rec.get("vs_f64_restatement_max", 0.0), er)
        assert t.pos() == pos + 1
        if gen:
            r32 = ref.astype(np.float32)
            want = int(np.argmax(r32))
            got_tok = int(t.token(pos + 1))
            if got_tok != want:
                top2 = np.sort(r32)[-2:]
                assert top2[1] - top2[0] <= 4e-3 * max(1.0, abs(top2[1])), "pos %d: token %d vs %d without a near-tie" % (pos, got_tok, want)
                ties += 1
                ring[pos + 1] = want                      # keep both evaluations on the restatement's sequence
            compared += 1
            toks.append(want)
    if rec is not None:
        rec.update({"logits_max_rel_vs_restatement": worst, "tokens_compared": compared, "near_ties": ties, "positions": steps,
                    "per_position_vs_restatement": per_pos, "bound": bound_vs_restatement})
    return worst, compared, ties


def test_config2_llama2_7b_decode_32_positions(q4, orc, m7b, observed):
    t = q4.Transformer(m7b)
    m = orc.Model(m7b)
    worst, compared, ties = _lockstep(q4, t, m, PROMPT, 32, 32, bound_vs_restatement=BOUND["7b"],
                                     rec=observed.setdefault("config2_7b", {}))
    assert compared == 25 and ties <= 2
    # KV rows of the last position: layer 0 sees only the embedding (one GEMV deep), the last layer the whole stack
    rk, rv = m.kv()
    for layer, tol in ((0, 2e-3), (t.config.n_layers - 1, 0.1)):
        gk, gv = t.kv_row(layer, 31)
        assert _rel(gk, rk[layer, 31]) <= tol and _rel(gv, rv[layer, 31]) <= tol, layer
    t.close()
    m.close()


def test_contract_tolerance_at_full_depth_on_a_contractive_7b(q4, orc, observed):
    """SURVEY section 8(c)'s contract figure -- logits within 3e-2 * max(1, |logit|) of the reference-order restatement -- exercised at FULL depth (VERDICT r05
    item 2a). The random-weight 7B / 13B models above are chaotic by construction (a rounding that falls the other way is amplified by every layer behind
    it), so their bound is the f64 bracket; `7b_contractive` is the same shape, the same 32 layers and seeds with the scales of the residual-writing
    matrices (o, down) x 0.25: what remains between two fp16 evaluations is what the arithmetic itself contributes. 32 positions in lockstep, every fusion
    level (5: the FFN half + the next layer's QKV as one launch; 4: the FFN half as one launch; 3; 1; 0: the reference's 1:1 launch list), greedy tokens exact outside near-ties."""
    path = _model("7b_contractive")
    L = q4.lib()
    m = orc.Model(path)
    refs, want = [], []
    toks = list(PROMPT)
    rec = observed.setdefault("contract_7b_contractive", {})
    try:
        for level in (3, 5, 4, 1, 0):
            L.q4_set_fusion(level)
            t = q4.Transformer(path)
            t.reset(PROMPT)
            ring = _ring(t)
            worst, ties, per_pos = 0.0, 0, []
            for pos in range(32):
                gen = pos >= len(PROMPT) - 1
                t.run_transformer(gen)
                q4.synchronize()
                if len(refs) <= pos:                      # the restatement runs once; every level is fed ITS tokens
                    refs.append(m.forward(toks[pos], pos).copy())
                    if gen:
                        want.append(int(np.argmax(refs[pos].astype(np.float32))))
                        toks.append(want[-1])
                e = _rel(t.logits(), refs[pos])
                worst = max(worst, e)
                per_pos.append(round(e, 5))
                assert e <= 3e-2, "fusion %d pos %d: max rel logit err %g vs the restatement (contract: 3e-2)" % (level, pos, e)
                if gen:
                    w = want[pos - (len(PROMPT) - 1)]
                    if int(t.token(pos + 1)) != w:
                        top2 = np.sort(refs[pos].astype(np.float32))[-2:]
                        assert top2[1] - top2[0] <= 4e-3 * max(1.0, abs(top2[1])), "fusion %d pos %d: token without a near-tie" % (level, pos)
                        ties += 1
                        ring[pos + 1] = w
            q4.check(L.q4_handoff_status(t.state))
            rec["fusion%d" % level] = {"logits_max_rel_vs_restatement": worst, "near_ties": ties, "positions": 32, "bound": 3e-2, "per_position": per_pos}
            assert ties <= 2
            t.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
        m.close()


def test_config3_llama2_13b_decode_24_positions(q4, orc, observed):
    path = _model("13b")
    t = q4.Transformer(path)
    m = orc.Model(path)
    worst, compared, ties = _lockstep(q4, t, m, PROMPT, 24, 24, bound_vs_restatement=BOUND["13b"], rec=observed.setdefault("config3_13b", {}))
    assert compared == 17 and ties <= 2
    t.close()
    m.close()


def test_grouped_query_model_at_mistral_7b_geometry(q4, orc, observed):
    path = _model("mistral7b")
    t = q4.Transformer(path)
    m = orc.Model(path)
    worst, compared, ties = _lockstep(q4, t, m, PROMPT, 24, 8, bound_vs_restatement=BOUND["mistral7b"], rec=observed.setdefault("mistral7b_gqa", {}))
    assert compared == 17 and ties <= 2
    t.close()
    m.close()


def test_mistral_geometry_fused_equals_unfused_bits(q4):
    """Mistral geometry: hidden_dim 14336 runs the gate/up GEMV as strips (csrc/gemv_strip.h) and dim 4096 x vocabulary 32000 the classifier
    (gemv_strip_cls.h). At fusion level 1 the rmsnorms are fused into the strips' x staging (the final one into the classifier's), at level 0
    they are launches of their own: the two launch sequences must leave identical logits (same canonical reductions, same rounding
    points), as they do for the wave-owned kernels (test_forward_gpu::test_fused_equals_unfused_bits; level 3 sums an attention output's
    positions in another fp32 order below bin 512 -- DESIGN.md section 3.3 -- and is held to the restatement by the lockstep tests above.
    At K = 5120 -- 13B -- levels 0 and 1 differ with the wave-owned kernels already: level 0's three q / k / v launches and level 1's fused
    one give the shared half slot to different lanes; there the strips are pinned to the wave-owned kernels bit for bit in prof_cases.py)."""
    L = q4.lib()
    path = _model("mistral7b")
    outs = []
    try:
        for fusion in (0, 1):
            L.q4_set_fusion(fusion)
            t = q4.Transformer(path)
            t.reset([1, 5, 9])
            for pos in range(6):
                t.run_transformer(pos >= 2)
            q4.synchronize()
            outs.append(t.logits().copy())
            t.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))


def _kv_to_host(q4, t):
    cfg = t.config
    kv_dim = cfg.dim * cfg.n_kv_heads // cfg.n_heads
    n = cfg.n_layers * cfg.seq_len * kv_dim
    k = np.empty(n, dtype=np.uint16)
    v = np.empty(n, dtype=np.uint16)
    L = q4.lib()
    q4.check(L.q4_memcpy_d2h(k.ctypes.data, t.state.contents.key_cache, k.nbytes))
    q4.check(L.q4_memcpy_d2h(v.ctypes.data, t.state.contents.value_cache, v.nbytes))
    return k, v


def _step_from_the_gpus_cache(q4, orc, path, target, observed, key, bound=BOUND["7b"]):
    """Decode `target` positions through run_transformer's captured graphs (every sequence-length bin up to 2048: bin 256 with
    four attention blocks per head, one V slice each, the split-context attention from bin 512 on), then compare ONE more
    step with the restatement started from the GPU's own KV cache -- no 2000-step CPU run, and every cached position takes
    part in the compared step's attention."""
    L = q4.lib()
    assert L.q4_get_fusion() == q4.DEFAULT_FUSION
    t = q4.Transformer(path)
    m = orc.Model(path)
    toks, tps, timed, _ = t.generate_ids(PROMPT, target)          # positions 0 .. target-1 (graphs of 128 .. 2048)
    assert timed == target - 1 and t.pos() == target
    k, v = _kv_to_host(q4, t)
    n = k.shape[0]
    np.ctypeslib.as_array(m.L.orc_key_cache(m.h), shape=(n,))[:] = k
    np.ctypeslib.as_array(m.L.orc_value_cache(m.h), shape=(n,))[:] = v
    tok = int(t.token(target))
    t.run_transformer(True)                                       # position `target`, graph of its bin (256 / 512 / 1024 / 2048)
    q4.synchronize()
    got = t.logits()
    ref = m.forward(tok, target)
    assert t.pos() == target + 1
    # layer 0's K/V row depends on the embedding only; layer 1's on layer 0's attention over all `target` cached positions
    rk, rv = m.kv()
    rec = observed.setdefault("%s_pos%d" % (key, target), {"tok_s_to_target": tps})
    # K rows carry RoPE at a large angle (two 1-ulp GEMV outputs rotated: up to 4 fp16 ulps), V rows are one GEMV deep
    for layer, ktol, vtol in ((0, 8e-3, 3e-3), (1, 2.4e-2, 1.2e-2), (t.config.n_layers - 1, 0.1, 0.1)):
        gk, gv = t.kv_row(layer, target)
        ek, ev = _rel(gk, rk[layer, target]), _rel(gv, rv[layer, target])
        rec["layer%d_k_max_rel" % layer], rec["layer%d_v_max_rel" % layer] = ek, ev
        assert ek <= ktol and ev <= vtol, (layer, ek, ev)
    e = _rel(got, ref)
    rec["logits_max_rel_vs_restatement"] = e
    rec["bound"] = bound
    assert e <= bound, e
    r32 = ref.astype(np.float32)
    top2 = np.sort(r32)[-2:]
    if top2[1] - top2[0] > 4e-3 * max(1.0, abs(top2[1])):
        assert int(t.token(target + 1)) == int(np.argmax(r32))
    t.close()
    m.close()


@pytest.mark.parametrize("target", [200, 400, 900, 1100, 2040])
def test_config4_llama2_7b_long_context_through_captured_graphs(q4, orc, m7b, target, observed):
    _step_from_the_gpus_cache(q4, orc, m7b, target, observed, "config4_7b")


@pytest.mark.parametrize("name", ["long16k", "long16k_h128"])
@pytest.mark.parametrize("fusion", [3, 1])
@pytest.mark.parametrize("target", [3000, 6000, 12000])
def test_bins_above_2048_end_to_end(q4, orc, tmp_path_factory, observed, name, fusion, target):
    """run_transformer's graph bins above 2048 (llama2_q4.cu:356-360): position 3000 -> bin 4096, 6000 -> bin 8192, 12000 -> the
    LAST bin, which holds the model's seq_len (16384: the reference's attention then runs softmax_kernel_no_smem, :276-279, which
    the restatement follows). The GPU decodes `target` positions through its captured graphs, the restatement takes the GPU's KV
    cache and both run the next step. Head 64 and head 128, grouped-query; fusion level 3 (attention -> o-proj launch where the
    shape has a form, split-context records as granules) and level 1 (stand-alone kernels)."""
    L = q4.lib()
    path = str(tmp_path_factory.getbasetemp() / (name + "_seed11.bin"))
    if not os.path.exists(path):
        synth.write_model(path, name, seed=11)
    try:
        L.q4_set_fusion(fusion)
        t = q4.Transformer(path)
        m = orc.Model(path)
        assert orc.lib().orc_seq_len_bin(target, 16384) == {3000: 4096, 6000: 8192, 12000: 16384}[target]
        toks, tps, timed, _ = t.generate_ids(PROMPT_SMALL, target)
        assert timed == target - 1 and t.pos() == target
        k, v = _kv_to_host(q4, t)
        n = k.shape[0]
        np.ctypeslib.as_array(m.L.orc_key_cache(m.h), shape=(n,))[:] = k
        np.ctypeslib.as_array(m.L.orc_value_cache(m.h), shape=(n,))[:] = v
        tok = int(t.token(target))
        t.run_transformer(True)
        q4.synchronize()
        q4.check(L.q4_handoff_status(t.state))
        got = t.logits()
        ref = m.forward(tok, target)
        e = _rel(got, ref)
        observed["%s_fusion%d_pos%d" % (name, fusion, target)] = {"logits_max_rel_vs_restatement": e, "tok_s_to_target": tps}
        assert e <= 3e-3, e            # two layers deep: measured <= 1.0e-3, one fp16 ulp of an O(1) logit (profiles/r04_parity_observed.json)
        rk, rv = m.kv()
        for layer in range(t.config.n_layers):
            gk, gv = t.kv_row(layer, target)
            assert _rel(gk, rk[layer, target]) <= 2.4e-2 and _rel(gv, rv[layer, target]) <= 1.2e-2, layer
        r32 = ref.astype(np.float32)
        top2 = np.sort(r32)[-2:]
        if top2[1] - top2[0] > 4e-3 * max(1.0, abs(top2[1])):
            assert int(t.token(target + 1)) == int(np.argmax(r32))
        t.close()
        m.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)


def test_llama2_13b_inside_bin_256(q4, orc, observed):
    """The same comparison at the 13B geometry inside bin 256 (K = 5120 in three k-slots with a shared half slot, 40 heads x
    4 V-slice attention blocks, 160 o-proj blocks)."""
    _step_from_the_gpus_cache(q4, orc, _model("13b"), 200, observed, "config3_13b", bound=BOUND["13b"])


def test_llama2_13b_inside_bin_1024(q4, orc, observed):
    """... and inside bin 1024 (split-context attention: 40 heads x eight 128-position chunks, records as granules, merged by each
    head's first chunk block), position 600."""
    _step_from_the_gpus_cache(q4, orc, _model("13b"), 600, observed, "config3_13b", bound=BOUND["13b"])


def test_config5_llama2_7b_perplexity_path_64_positions(q4, orc, m7b, observed):
    npos = 64
    rng = np.random.default_rng(5)
    toks = np.concatenate([[1], rng.integers(3, 32000, size=npos)]).astype(np.int32)
    t = q4.Transformer(m7b, perplexity=True)
    m = orc.Model(m7b)
    ppl = t.perplexity_ids(toks)                                   # get_dataset_perplexity perplexity.h:57-97 on ids
    glog = t.logits_array(npos)
    rlog = np.stack([m.forward(int(toks[i]), i).astype(np.float32) for i in range(npos)])
    n64 = 8
    for i in range(n64):
        exact = m.forward_f64(int(toks[i]), i, cap=n64)
        eg, er = _rel(glog[i], exact), _rel(rlog[i], exact)
        assert eg <= 2.0 * er + 1e-3, (i, eg, er)
        assert _rms(glog[i], exact) <= 1.5 * _rms(rlog[i], exact) + 1e-4, (i, _rms(glog[i], exact), _rms(rlog[i], exact))
    assert _rel(glog, rlog) <= BOUND["7b"]
    rppl = orc.compute_perplexity(toks[1:npos + 1], rlog)
    observed["config5_7b_perplexity"] = {"positions": npos, "perplexity_gpu": float(ppl), "perplexity_restatement": float(rppl),
                                         "logits_max_rel_vs_restatement": _rel(glog, rlog)}
    assert abs(ppl - rppl) <= 5e-3 * rppl, (ppl, rppl)            # perplexity within 0.5 %
    t.close()
    m.close()


def test_repeated_generations_are_identical_and_no_handoff_times_out(q4, m7b):
    """Soak of the default path (attention -> o-proj as one launch, in-launch hand-off): 50 `-n 256` generations must reproduce
    the first one's token ring, leave the hand-off error word clear and the library at fusion level 3."""
    L = q4.lib()
    assert L.q4_get_fusion() == q4.DEFAULT_FUSION
    before = L.q4_handoff_timeouts()
    t = q4.Transformer(m7b)
    ref = t.generate_ids(PROMPT, 256)[0].copy()
    for run in range(50):
        toks = t.generate_ids(PROMPT, 256)[0]
        assert np.array_equal(toks, ref), "token ring changed in run %d" % run
    q4.check(L.q4_handoff_status(t.state))
    assert L.q4_handoff_timeouts() == before and L.q4_get_fusion() == q4.DEFAULT_FUSION
    t.close()


def test_repeated_long_generations_through_the_split_context_bins(q4, m7b):
    """The same soak through every sequence-length bin: 8 `-n 2048` generations (partial records of the split-context attention
    cross CUs as data-tagged granules, merged by each head's first chunk block) must reproduce the first token ring."""
    L = q4.lib()
    before = L.q4_handoff_timeouts()
    t = q4.Transformer(m7b)
    ref = t.generate_ids(PROMPT, 2048)[0].copy()
    for run in range(8):
        toks = t.generate_ids(PROMPT, 2048)[0]
        assert np.array_equal(toks, ref), "token ring changed in run %d" % run
    q4.check(L.q4_handoff_status(t.state))
    assert L.q4_handoff_timeouts() == before and L.q4_get_fusion() == q4.DEFAULT_FUSION
    t.close()


@pytest.mark.parametrize("n_cus", [32, 8])
def test_masked_stream_falls_back_to_the_launch_sequence(q4, m7b, n_cus):
    """On a stream restricted to a few CUs the blocks of the attention -> o-proj launch are not all resident at once: the
    residency guard (layer_attn.hip attention_oproj_form) must then run the stand-alone launches -- identical bits to fusion
    level 1 on the full device (first bin: same shapes) and no in-launch wait at all."""
    L = q4.lib()
    full = q4.lib().q4_get_stream()
    outs = {}
    try:
        for name, masked in (("level1_full", False), ("level3_masked", True)):
            s = C.c_void_p()
            if masked:
                q4.check(L.q4_stream_create_masked(C.byref(s), n_cus))
                L.q4_set_stream(s)
            L.q4_set_fusion(q4.DEFAULT_FUSION if masked else 1)
            t = q4.Transformer(m7b)
            t.reset(PROMPT)
            for pos in range(12):
                t.run_transformer(pos >= len(PROMPT) - 1)
            q4.synchronize()
            outs[name] = t.logits().view(np.uint16).copy()
            q4.check(L.q4_handoff_status(t.state))
            t.close()
            if masked:
                L.q4_set_stream(full)
                q4.check(L.q4_stream_destroy(s))
    finally:
        L.q4_set_stream(full)
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    # 8 CUs x 2 resident blocks < 160 blocks; at 32 CUs 64 < 160 as well: both run the launch sequence
    assert np.array_equal(outs["level1_full"], outs["level3_masked"])
