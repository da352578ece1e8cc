"""Cases that need the measurement knobs of the -DQ4_PROFILING build (libllama2_q4_prof.so): alternative issue orders and
slot forms of the int4 GEMV must not change results. Not collected by the default run (file name); run by
tests/test_profiling_build_gpu.py in a subprocess with Q4_PROFILING_BUILD=1, so the product library keeps exporting only
what include/llama2_q4.h declares."""
import numpy as np
import pytest

from conftest import assert_close_f16
from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [11008, 14336, 1024 * 4 + 8, 4096])
def test_gate_up_engine_equals_the_wave_owned_kernel(q4, rng, N):
    """The fused gate/up GEMV at K = 4096 has two forms (knob 11): -1 = gemv_q4_kernel<MODE_FFN> (the wave-owned kernel), 8 .. 14 = "strips"
    (csrc/gemv_strip.h: sixteen self-loading waves per CU on LDS-DMA rings; ring depth 2 / 4 / 8, plain and paced issue: csrc/exp/ffn_strip_variants.h);
    0 = the product's choice (strips from 36 columns per CU on, i.e. for 11008 and 14336 here). Same arithmetic in
    the same order: bit equality for the 7B and the Mistral hidden sizes, a ragged split over the CUs and the smallest covered width,
    repeated launches (a race between a fill and a read would show as a run-to-run difference)."""
    L = q4.lib()
    K = 4096
    x = rng.standard_normal(K).astype(np.float16)
    g, u = synth.random_qweight(rng, K, N), synth.random_qweight(rng, K, N)
    dg, du, dx = q4.DevQWeight(*g), q4.DevQWeight(*u), q4.DevBuf(x)
    outs = {}
    try:
        for engine in (-1, 0, 8, 9, 10, 12, 13, 14):
            L.q4_set_gemv_early(11, engine)
            for rep in range(6):
                dout = q4.DevBuf(nbytes=N * 2)
                q4.ffn_matvec_silu(dout, dx, dg, du, K, N)
                q4.synchronize()
                outs[(engine, rep)] = dout.get(np.float16, N).view(np.uint16).copy()
    finally:
        L.q4_set_gemv_early(11, 0)
    assert np.isfinite(outs[(-1, 0)].view(np.float16).astype(np.float32)).all()
    for key, o in outs.items():
        assert np.array_equal(o, outs[(-1, 0)]), key


@pytest.mark.parametrize("N", [13824, 13824 + 8, 12800, 5120, 14336, 8192])
def test_gate_up_strips_at_k5120_equal_the_shared_half_slot_kernel(q4, rng, N):
    """K = 5120 (13B): a column is two 1 KiB pieces and a half one. The strips form (csrc/gemv_strip.h, TS = 3) gives the half piece to the
    lower half of the wave for even columns and to the upper half for odd ones and adds its term as a product and a sum -- the lanes and
    the operations of gemv_q4.h's shared half slot -- so the two forms must agree bit for bit: knob 11 = -1 (wave-owned kernel), 0 (the
    product's choice: strips from 36 columns per CU on), 8 / 9 (strips, ring depth 2 / 4, wherever the shape is covered)."""
    L = q4.lib()
    K = 5120
    x = rng.standard_normal(K).astype(np.float16)
    g, u = synth.random_qweight(rng, K, N), synth.random_qweight(rng, K, N)
    dg, du, dx = q4.DevQWeight(*g), q4.DevQWeight(*u), q4.DevBuf(x)
    outs = {}
    try:
        for engine in (-1, 0, 8, 9, 19):     # 19: column units (two pieces and a half one) where 0 / 8 run pair units (five full pieces per column pair)
            L.q4_set_gemv_early(11, engine)
            for rep in range(4):
                dout = q4.DevBuf(nbytes=N * 2)
                q4.ffn_matvec_silu(dout, dx, dg, du, K, N)
                q4.synchronize()
                outs[(engine, rep)] = dout.get(np.float16, N).view(np.uint16).copy()
    finally:
        L.q4_set_gemv_early(11, 0)
    assert np.isfinite(outs[(-1, 0)].view(np.float16).astype(np.float32)).all()
    for key, o in outs.items():
        assert np.array_equal(o, outs[(-1, 0)]), key


@pytest.mark.parametrize("K,N", [(13824, 5120), (13824, 5120 + 8), (13824, 2048), (13824, 6144), (11008, 4096), (11008, 4096 + 8), (11008, 2048), (11008, 6144)])
@pytest.mark.parametrize("accum", [False, True])
def test_down_projection_strips_equal_the_k_split_kernel(q4, rng, K, N, accum):
    """The down projections (13B: K = 13824, four slots per k-part, the last shared between the columns of a pair; 7B: K = 11008, three slots,
    an ordinary last one): csrc/gemv_strip_down.h runs mat_vec_kernel_int4 as one sixteen-wave block per CU whose waves stream (column, k-part)
    units through LDS-DMA rings; the parts, the last slot's lanes and the part-0-then-part-1 sum are gemv_q4.h's K-split kernel's, so the two
    must agree bit for bit, with and without the residual add: knob 11 = -1 (K-split kernel) against 0 (the product: strips at K = 13824, where
    they are faster; the K-split kernel at K = 11008, where they are not) and 8 (strips wherever the shape is covered), for the model's width, a
    ragged split, the narrowest and a wide covered grid."""
    L = q4.lib()
    x = (rng.standard_normal(K) * 0.5).astype(np.float16)
    w = synth.random_qweight(rng, K, N)
    res = rng.standard_normal(N).astype(np.float16)
    dw, dx = q4.DevQWeight(*w), q4.DevBuf(x)
    outs = {}
    try:
        for engine in (-1, 0, 8):
            L.q4_set_gemv_early(11, engine)
            for rep in range(4):
                dout = q4.DevBuf(res)
                q4.matmul_q4(dout, dx, dw, K, N, accum=accum)
                q4.synchronize()
                outs[(engine, rep)] = dout.get(np.float16, N).view(np.uint16).copy()
    finally:
        L.q4_set_gemv_early(11, 0)
    assert np.isfinite(outs[(-1, 0)].view(np.float16).astype(np.float32)).all()
    for key, o in outs.items():
        assert np.array_equal(o, outs[(-1, 0)]), key


@pytest.mark.parametrize("n,d", [(4096, 32000), (5120, 32000), (4096, 32008), (5120, 16384), (4096, 128256)])
def test_classifier_strips_equal_gemv_f16_kernel(q4, rng, n, d):
    """The fp16 classifier GEMV as strips (csrc/gemv_strip_cls.h: one 16-wave block per CU, every wave streams whole vocabulary rows through its
    LDS-DMA ring) repeats gemv_f16_kernel's per-lane sums and its wave reduction: bit equality, knob 11 = -1 (gemv_f16_kernel) against 0 (the
    product: strips), at the Llama-2 sizes, a ragged split of rows, the narrowest covered matrix and the Llama-3 vocabulary."""
    L = q4.lib()
    w = (rng.standard_normal(n * d) * 0.02).astype(np.float16)
    x = rng.standard_normal(n).astype(np.float16)
    dw, dx = q4.DevBuf(w), q4.DevBuf(x)
    outs = {}
    try:
        for engine in (-1, 0):
            L.q4_set_gemv_early(11, engine)
            for rep in range(3):
                do = q4.DevBuf(nbytes=d * 2)
                q4.matmul(do, dx, dw, n, d)
                q4.synchronize()
                outs[(engine, rep)] = do.get(np.float16, d).view(np.uint16).copy()
    finally:
        L.q4_set_gemv_early(11, 0)
    assert np.isfinite(outs[(-1, 0)].view(np.float16).astype(np.float32)).all()
    for key, o in outs.items():
        assert np.array_equal(o, outs[(-1, 0)]), key


@pytest.mark.parametrize("K,N,kind", [(4096, 4096, 0), (11008, 4096, 1), (13824, 5120, 1), (4096, 11008, 3), (5120, 13824, 3)])
def test_early_bird_issue_order_is_bit_neutral(q4, rng, K, N, kind):
    """The early-bird issue order (first block per CU sends its weight loads before the staging completes) only moves
    loads in time: outputs must be bit-identical with it switched off, with the default, and with odd settings."""
    L = q4.lib()
    x = rng.standard_normal(K).astype(np.float16)
    dx, dout = q4.DevBuf(x), q4.DevBuf(nbytes=N * 2)
    if kind == 3:
        g, u = synth.random_qweight(rng, K, N), synth.random_qweight(rng, K, N)
        dg, du = q4.DevQWeight(*g), q4.DevQWeight(*u)
        run = lambda: q4.ffn_matvec_silu(dout, dx, dg, du, K, N)
    else:
        dw = q4.DevQWeight(*synth.random_qweight(rng, K, N))
        run = lambda: q4.matmul_q4(dout, dx, dw, K, N)
    outs = []
    try:
        for early in (0, 4, 4 | (8 << 8), 7, 64):
            L.q4_set_gemv_early(kind, early)
            run()
            q4.synchronize()
            outs.append(dout.get(np.float16, N).view(np.uint16).copy())
    finally:
        L.q4_set_gemv_early(kind, 4)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("K,N,kind", [(5120, 5120, "plain"), (5120, 13824, "ffn"), (5120, 5120, "qkv")])
def test_shared_half_slot_matches_the_full_slot_form(q4, orc, rng, K, N, kind):
    """K = 5120 leaves the third k-slot half empty; by default lanes 32-63 serve the pair's second column there (HALF).
    Both forms must agree with the oracle, and with each other to 1 fp16 ulp (the fp32 summation order differs)."""
    L = q4.lib()
    x = rng.standard_normal(K).astype(np.float16)
    dx = q4.DevBuf(x)
    outs = []
    try:
        if kind == "qkv":
            seq, pos = 4, 2
            mats = [synth.random_qweight(rng, K, N) for _ in range(3)]
            dws = [q4.DevQWeight(*m) for m in mats]
            for on in (0, 1):
                L.q4_set_half_tail(on)
                dq, dk, dv = q4.DevBuf(nbytes=N * 2), q4.DevBuf(nbytes=seq * N * 2), q4.DevBuf(nbytes=seq * N * 2)
                dpos = q4.DevBuf(np.array([pos], dtype=np.int32))
                q4.qkv_matvec(dq, dk, dv, dx, dws[0], dws[1], dws[2], K, N, 0, dpos)
                q4.synchronize()
                outs.append(np.concatenate([dq.get(np.float16, N), dk.get(np.float16)[pos * N:(pos + 1) * N], dv.get(np.float16)[pos * N:(pos + 1) * N]]))
            ref = np.concatenate([orc.matmul_q4(x, *m, K, N) for m in mats])
        else:
            g = synth.random_qweight(rng, K, N)
            u = synth.random_qweight(rng, K, N)
            dg, du, dout = q4.DevQWeight(*g), q4.DevQWeight(*u), q4.DevBuf(nbytes=N * 2)
            for on in (0, 1):
                L.q4_set_half_tail(on)
                if kind == "ffn":
                    q4.ffn_matvec_silu(dout, dx, dg, du, K, N)
                else:
                    q4.matmul_q4(dout, dx, dg, K, N)
                q4.synchronize()
                outs.append(dout.get(np.float16, N).copy())
            ref = orc.ffn_matvec_silu(x, g, u, K, N) if kind == "ffn" else orc.matmul_q4(x, *g, K, N)
    finally:
        L.q4_set_half_tail(1)
    for o in outs:
        assert_close_f16(o, ref, max_ulp=2 if kind == "ffn" else 1, max_frac=0.10, what="%s half-slot" % kind)
    assert_close_f16(outs[0], outs[1], max_ulp=2 if kind == "ffn" else 1, max_frac=0.10, what="%s half vs full" % kind)


def _model_file(name):
    import os
    geom = synth.GEOMETRIES[name]
    path = os.path.join(os.environ.get("Q4_MODEL_DIR", "/tmp"), "llama2_q4_synth_%s_seed20240229.bin" % name)
    if not (os.path.exists(path) and os.path.getsize(path) == synth.model_bytes(geom)):
        synth.write_model(path, geom)
    return path


@pytest.mark.parametrize("model,n_cus,steps", [("7b", 32, 48), ("7b", 8, 48), ("head128", 8, 1060), ("head128", 3, 700)])
def test_handoff_makes_progress_with_fewer_resident_blocks_than_the_grid(q4, model, n_cus, steps):
    """The attention -> o-proj launch on a CU-masked stream with the residency guard switched OFF (a knob of the profiling
    build): 160-400 blocks on 3 / 8 / 32 CUs, so most o-proj blocks are dispatched only after earlier blocks have left. The
    product never runs it that way (the guard falls back to the launch sequence, tests/test_baseline_configs_gpu.py); this
    case shows what the guard protects against does not bite on this hardware either: work-groups are dispatched in index
    order, producers come first and wait for nobody. No bounded poll may run out; logits within the model's bound of fusion
    level 1 (split-context bins included for the small model)."""
    import ctypes as C
    L = q4.lib()
    full = L.q4_get_stream()
    path = _model_file(model)
    outs = {}
    cps = (3, 40, 47) if model == "7b" else (3, 127, 255, 511, 600, steps - 1)
    try:
        for lvl in (1, 3):
            s = C.c_void_p()
            q4.check(L.q4_stream_create_masked(C.byref(s), n_cus))
            L.q4_set_stream(s)
            L.q4_set_gemv_early(8, 0)          # guard off
            L.q4_set_fusion(lvl)
            t = q4.Transformer(path)
            t.reset([1, 5, 9])
            got = []
            every = {}
            for pos in range(steps):
                t.run_transformer(pos >= 2)
                if pos in cps or (model == "7b" and pos >= 2):
                    q4.synchronize()
                    every[pos] = t.logits().copy()
                    if pos in cps:
                        got.append(every[pos])
            q4.check(L.q4_handoff_status(t.state))
            assert L.q4_handoff_timeouts() == 0 and L.q4_get_fusion() == lvl
            outs[lvl] = (got, [int(t.token(i)) for i in range(steps + 1)], every)
            t.close()
            L.q4_set_stream(full)
            q4.check(L.q4_stream_destroy(s))
    finally:
        L.q4_set_stream(full)
        L.q4_set_gemv_early(8, 1)
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    for a, b, pos in zip(outs[1][0], outs[3][0], cps):
        if outs[1][1][:pos + 1] != outs[3][1][:pos + 1]:
            # the two levels are two valid fp32 groupings: their greedy rings may part, but only where the two best logits nearly tie
            first = next(i for i, (x, y) in enumerate(zip(outs[1][1], outs[3][1])) if x != y)      # token first written by step first - 1
            assert first - 1 in outs[1][2], "token rings diverged at %d (no logits kept there)" % first
            top2 = np.sort(outs[1][2][first - 1].astype(np.float32))[-2:]
            assert top2[1] - top2[0] <= 4e-3 * max(1.0, abs(top2[1])), "token rings diverged at %d without a near-tie (%g vs %g)" % (first, top2[1], top2[0])
            break
        af, bf = a.astype(np.float64), b.astype(np.float64)      # (another fp32 grouping of the positions: the model's bound;
        err = float((np.abs(af - bf) / np.maximum(1.0, np.abs(bf))).max())   # 32 layers of the 7B geometry amplify a few ulps)
        assert err <= (5e-2 if model == "7b" else 5e-3 if pos <= 128 else 1.2e-2), (pos, err)


def test_a_timed_out_handoff_is_reported_once_and_the_sequence_is_redone_at_level_1(q4, tmp_path):
    """A REAL time-out: the attention blocks of one launch do not publish (profiling knob), the o-proj blocks' bounded polls
    run out and set the model's error word; launches queued behind it do not spin again (`dead`). q4_generate_ids must notice,
    clear the state, drop to fusion level 1 and return the tokens of a clean run; q4_perplexity_ids likewise."""
    L = q4.lib()
    p = str(tmp_path / "head128.bin")
    synth.write_model(p, "head128", seed=7)
    prompt = [1, 5, 9]
    try:
        L.q4_set_fusion(1)
        t = q4.Transformer(p, perplexity=True)
        want = t.generate_ids(prompt, 40)[0].copy()
        ptoks = np.concatenate([[1], np.arange(3, 23)]).astype(np.int32)
        want_ppl = t.perplexity_ids(ptoks)
        L.q4_set_fusion(q4.DEFAULT_FUSION)
        assert np.array_equal(t.generate_ids(prompt, 40)[0][:12], want[:12])   # level 3, nothing sabotaged: same greedy tokens
        assert L.q4_handoff_timeouts() == 0
        L.q4_set_gemv_early(9, 1)                                       # the next attention -> o-proj launch is captured mute
        got = t.generate_ids(prompt, 40)[0]
        assert np.array_equal(got, want)
        assert L.q4_handoff_timeouts() == 1 and L.q4_get_fusion() == 1
        q4.check(L.q4_handoff_status(t.state))                          # reported once, state clean now
        L.q4_set_fusion(q4.DEFAULT_FUSION)
        L.q4_set_gemv_early(9, 1)
        ppl = t.perplexity_ids(ptoks)
        assert ppl == want_ppl and L.q4_handoff_timeouts() == 2 and L.q4_get_fusion() == 1
        # probation: the second time-out sits out 32 sequences (16 after the first, doubled), counted by q4_reset_sequence -- the
        # redo of the failed sequence was the first of them --, then level 3 is tried again by itself, with clean results
        for i in range(30):
            assert np.array_equal(t.generate_ids(prompt, 12)[0], want[:13]) and L.q4_get_fusion() == 1, i
        assert np.array_equal(t.generate_ids(prompt, 40)[0][:12], want[:12])
        assert L.q4_get_fusion() == q4.DEFAULT_FUSION and L.q4_handoff_timeouts() == 2
        t.close()
    finally:
        L.q4_set_gemv_early(9, 0)
        L.q4_set_fusion(q4.DEFAULT_FUSION)


@pytest.mark.parametrize("model,steps", [("head128", 300), ("head128_k5120", 300), ("head128_gqa", 300), ("tinyllama", 300), ("head256", 300)])
def test_one_block_and_v_slice_forms_of_the_attention_role(q4, model, steps):
    """Attention role of the attention -> o-proj launch below the split-context bins. Knob 10 = 0: one block per head, the
    stand-alone kernel's body -- for head 128 in the stand-alone kernel's shape, so in the first bin fusion level 3 must equal
    level 1 BIT FOR BIT (any stale or torn granule of the hand-off would show). Knob 10 = 1, the default: head_size / 32 blocks
    per head, every one computes all scores and the softmax and takes one 64-byte slice of the V rows, 16 positions per wave
    instruction: same scores, same rounding points, another fp32 order over the positions -- the model's bound, equal greedy
    tokens until a near-tie."""
    L = q4.lib()
    path = _model_file(model)
    cps = (0, 3, 60, 127, 128, 200, 255, 256, steps - 1)
    outs = {}
    timeouts = L.q4_handoff_timeouts()      # (a counter of the process: an earlier case provokes a real one)
    try:
        for key, lvl, knob in (("level1", 1, 1), ("one_block", 3, 0), ("v_slices", 3, 1)):
            L.q4_set_gemv_early(10, knob)
            L.q4_set_fusion(lvl)
            t = q4.Transformer(path)
            t.reset([1, 5, 9])
            got = []
            for pos in range(steps):
                t.run_transformer(pos >= 2)
                if pos in cps:
                    q4.synchronize()
                    got.append(t.logits().view(np.uint16).copy())
            q4.check(L.q4_handoff_status(t.state))
            assert L.q4_handoff_timeouts() == timeouts and L.q4_get_fusion() == lvl
            outs[key] = (got, [int(t.token(i)) for i in range(steps + 1)])
            t.close()
    finally:
        L.q4_set_gemv_early(10, 1)
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    if model.startswith("head128"):
        assert outs["one_block"][1][:128] == outs["level1"][1][:128]
        for a, b, pos in zip(outs["level1"][0], outs["one_block"][0], cps):
            if pos < 128:
                assert np.array_equal(a, b), "one-block form differs from fusion level 1 at position %d" % pos
    for other in ("one_block", "level1"):
        for a, b, pos in zip(outs[other][0], outs["v_slices"][0], cps):
            if outs[other][1][:pos + 1] != outs["v_slices"][1][:pos + 1]:
                assert pos >= 60, "token rings diverged early (%d, V slices vs %s)" % (pos, other)
                break
            af, bf = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
            err = float((np.abs(af - bf) / np.maximum(1.0, np.abs(bf))).max())
            assert err <= (5e-3 if pos <= 128 else 1.2e-2), (pos, other, err)   # (histories differ by an ulp from early on; measured 5.4e-3 at 256)


@pytest.mark.parametrize("model,steps", [("head128", 1090), ("tinyllama", 1090)])
def test_held_back_oproj_weight_requests_are_bit_neutral(q4, model, steps):
    """Split-context bins (round 5): the o-proj role of the attention -> o-proj launch requests its weights only when the K / V stream is
    about to end (knob 15: per cent of the stream's estimated duration; -1 = the product's per-bin rule, 0 = at entry). Timing only: logits
    and token rings must be equal bit for bit whatever the hold, including one far longer than the stream (300 %)."""
    L = q4.lib()
    path = _model_file(model)
    cps = sorted({511, 513, 600, 1023, 1030, steps - 1} & set(range(steps)))
    outs = {}
    timeouts = L.q4_handoff_timeouts()
    try:
        for hold in (0, -1, 100, 300):
            L.q4_set_gemv_early(15, hold)
            t = q4.Transformer(path)
            t.reset([1, 5, 9])
            got = []
            for pos in range(steps):
                t.run_transformer(pos >= 2)
                if pos in cps:
                    q4.synchronize()
                    got.append(t.logits().view(np.uint16).copy())
            q4.check(L.q4_handoff_status(t.state))
            assert L.q4_handoff_timeouts() == timeouts
            outs[hold] = (got, [int(t.token(i)) for i in range(steps + 1)])
            t.close()
    finally:
        L.q4_set_gemv_early(15, -1)
    for hold in (-1, 100, 300):
        assert outs[hold][1] == outs[0][1], "token rings differ (hold %d)" % hold
        for a, b, pos in zip(outs[0][0], outs[hold][0], cps):
            assert np.array_equal(a, b), "hold %d: logits differ at position %d" % (hold, pos)


def test_ffn_pair_launch_times_out_cleanly_and_its_gather_modes_agree(q4, tmp_path):
    """The FFN half of a layer as one launch (csrc/gemv_ffn_pair.h, fusion level 4). (1) Its gather modes (profiling knob 16: first pass with sc1
    or plain loads, at once or when the wave's own down pieces have landed) are timing choices: token rings and logits identical. (2) A REAL
    time-out: the blocks of one launch do not publish (knob 17), every bounded wait runs out, the error word is set, launches queued behind it
    do not spin again; q4_generate_ids redoes the sequence at fusion level 1 with the tokens of a clean run; q4_set_fusion(level) re-arms the launch. Levels 4, 5 (with the next
    layer's QKV as the launch's third phase) and 6 (with this layer's attention and output projection in front: the attention units do not publish either)."""
    L = q4.lib()
    p = str(tmp_path / "ffn_pair7b.bin")
    synth.write_model(p, "ffn_pair7b", seed=11)
    prompt = [1, 5, 9]
    try:
        L.q4_set_fusion(1)
        t = q4.Transformer(p)
        want = t.generate_ids(prompt, 40)[0].copy()
        before = L.q4_handoff_timeouts()
        for level in (4, 5, 6):
            for mode in (3, 0, 1, 2, 19):
                L.q4_set_fusion(level)
                L.q4_set_gemv_early(16, mode)
                assert np.array_equal(t.generate_ids(prompt, 40)[0], want), (level, mode)      # bit-identical launches: the same greedy ring
                assert L.q4_handoff_timeouts() == before and L.q4_get_fusion() == level
        L.q4_set_gemv_early(16, 3)
        for k, level in enumerate((4, 5, 6)):
            L.q4_set_fusion(level)
            L.q4_set_gemv_early(17, 1)                                  # the next FFN pair launch is captured mute
            got = t.generate_ids(prompt, 40)[0]
            assert np.array_equal(got, want)
            assert L.q4_handoff_timeouts() == before + 1 + k and L.q4_get_fusion() == 1
            q4.check(L.q4_handoff_status(t.state))                      # reported once, state clean now
            assert np.array_equal(t.generate_ids(prompt, 12)[0], want[:13]) and L.q4_get_fusion() == 1     # probation (its length is process-wide state: see the test above)
            L.q4_set_fusion(level)                                      # re-armed at once by an explicit choice
            assert np.array_equal(t.generate_ids(prompt, 40)[0], want)
            assert L.q4_get_fusion() == level and L.q4_handoff_timeouts() == before + 1 + k
        t.close()
    finally:
        L.q4_set_gemv_early(17, 0)
        L.q4_set_gemv_early(16, 3)
        L.q4_set_fusion(q4.DEFAULT_FUSION)
