"""Temperature / top-p sampling (sampler.h:51-81, gpu_kernels.h:499-584) on the device vs the CPU restatement,
same coin from the same xorshift stream, including the fp16 rounding points and the fp16 prefix sum."""
import ctypes as C

import numpy as np
import pytest

from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("m") / "small.bin")
    synth.write_model(p, "small", seed=21)
    return p


@pytest.mark.parametrize("temperature,topp", [(0.5, 0.6), (1.0, 0.9), (0.8, 1.0), (1.3, 0.0), (0.3, 0.95),
                                              (0.05, 0.9), (0.12, 0.6)])   # peaked: the top entry alone reaches the threshold
def test_sample_matches_restatement(q4, orc, model, temperature, topp):
    L = q4.lib()
    t = q4.Transformer(model, temperature=temperature, topp=topp, seed=1234)
    vocab = t.config.vocab_size
    rng = np.random.default_rng(int(temperature * 100 + topp * 10))
    state = C.c_ulonglong(1234)
    mism = 0
    for trial in range(24):
        logits = (rng.standard_normal(vocab) * (1.0 + trial % 4)).astype(np.float16)
        if trial % 5 == 0:
            logits[rng.integers(0, vocab, 8)] = logits.max()           # ties in the sorted order
        t.reset([1])
        q4.check(L.q4_memcpy_h2d(t.state.contents.logits, logits.ctypes.data, logits.nbytes))
        q4.check(L.q4_sample(t.sampler, t.state, 1))
        q4.synchronize()
        coin = L.random_f32(C.byref(state))                             # the same stream the sampler advanced
        ref = orc.lib().orc_sample_topp(orc.f16_bits(logits.copy()), vocab, temperature, topp, coin)
        assert t.pos() == 1
        mism += int(t.token(1) != ref)
    assert mism == 0
    t.close()


@pytest.fixture(scope="module")
def big_vocab_models(tmp_path_factory):
    d = tmp_path_factory.mktemp("v")
    out = {}
    for name in ("v32k", "v40k"):
        out[name] = str(d / (name + ".bin"))
        synth.write_model(out[name], name, seed=5)
    return out


def _logit_case(rng, vocab, trial):
    """Distributions the sorted-prefix search must get right: ordinary, wide, flat (the fp16 prefix sum saturates),
    exactly flat, ties at the maximum, ties everywhere (few distinct values), one dominant entry."""
    kind = trial % 7
    if kind == 0:
        x = rng.standard_normal(vocab) * 2.0
    elif kind == 1:
        x = rng.standard_normal(vocab) * 6.0
    elif kind == 2:
        x = rng.standard_normal(vocab) * 0.05                   # nearly flat: ~1/vocab each
    elif kind == 3:
        x = np.full(vocab, float(rng.standard_normal()))         # exactly flat: every key equal
    elif kind == 4:
        x = rng.standard_normal(vocab) * 2.0
        x[rng.integers(0, vocab, 8)] = x.max() + 0.5             # 8-way tie at the top
    elif kind == 5:
        x = rng.integers(-3, 4, vocab).astype(np.float64)        # 7 distinct values: long runs of equal keys
    else:
        x = rng.standard_normal(vocab)
        x[int(rng.integers(0, vocab))] = 12.0
    return x.astype(np.float16)


@pytest.mark.parametrize("name", ["v32k", "v40k"])
@pytest.mark.parametrize("temperature,topp", [(0.5, 0.6), (1.0, 0.9), (0.8, 1.0), (1.3, 0.0), (0.3, 0.95), (0.05, 0.9),
                                              (0.12, 0.6)])
def test_sample_matches_restatement_at_production_vocab(q4, orc, big_vocab_models, name, temperature, topp):
    """sample() sampler.h:51-81 where the CLI runs it: vocab 32000 (32 register-resident keys per thread, packed ranks,
    transposed LDS layout) and vocab 40000 (> 32768: the same passes through global memory), 105 trials per (T, p)
    over ordinary / flat / tied distributions, token-exact against the restatement with the same coin stream."""
    L = q4.lib()
    t = q4.Transformer(big_vocab_models[name], temperature=temperature, topp=topp, seed=99)
    vocab = t.config.vocab_size
    rng = np.random.default_rng(int(temperature * 1000 + topp * 10) + vocab)
    state = C.c_ulonglong(99)
    bad = []
    for trial in range(105):
        logits = _logit_case(rng, vocab, trial)
        t.reset([1])
        q4.check(L.q4_memcpy_h2d(t.state.contents.logits, logits.ctypes.data, logits.nbytes))
        q4.check(L.q4_sample(t.sampler, t.state, 1))
        q4.synchronize()
        coin = L.random_f32(C.byref(state))
        ref = orc.lib().orc_sample_topp(orc.f16_bits(logits.copy()), vocab, temperature, topp, coin)
        assert t.pos() == 1
        if t.token(1) != ref:
            bad.append((trial, trial % 7, int(t.token(1)), ref, coin))
    assert not bad, bad[:8]
    t.close()


@pytest.mark.parametrize("temperature,topp", [(-0.7, 0.9), (-1.5, 1.0)])
def test_negative_temperature_through_the_api_stays_finite(q4, orc, big_vocab_models, temperature, topp):
    """build_sampler does not clamp the temperature (only the CLI does, llama2_q4.cu:682): with T < 0 the reference's softmax still
    takes the maximum of the SCALED logits (gpu_kernels.h:511,522-531) and stays finite. The 16-CU spread softmax derives that maximum
    from max(logit) / T, which holds for T > 0 only, so such a sampler goes the one-block way: token-exact against the restatement."""
    L = q4.lib()
    t = q4.Transformer(big_vocab_models["v32k"], temperature=temperature, topp=topp, seed=5)
    vocab = t.config.vocab_size
    rng = np.random.default_rng(77)
    state = C.c_ulonglong(5)
    bad = []
    for trial in range(21):
        logits = _logit_case(rng, vocab, trial)
        t.reset([1])
        q4.check(L.q4_memcpy_h2d(t.state.contents.logits, logits.ctypes.data, logits.nbytes))
        q4.check(L.q4_sample(t.sampler, t.state, 1))
        q4.synchronize()
        coin = L.random_f32(C.byref(state))
        ref = orc.lib().orc_sample_topp(orc.f16_bits(logits.copy()), vocab, temperature, topp, coin)
        if t.token(1) != ref:
            bad.append((trial, trial % 7, int(t.token(1)), ref, coin))
    assert not bad, bad[:8]
    t.close()


@pytest.mark.parametrize("temperature,topp", [(0.5, 0.6), (1.0, 0.9), (0.8, 1.0), (0.05, 0.9)])
def test_sampled_steps_inside_the_graph_equal_the_stepwise_sampler(q4, model, temperature, topp):
    """Sampled generation goes out eight steps per graph replay: topp_sample_kernel is part of the captured graph, reads its
    coin from the device ring by position (the host draws the same xorshift stream, sampler.h:31-45) and leaves the sampled
    token's embedding row for the next step. The token ring must equal the one of the reference-shaped loop -- one
    run_transformer call per step (llama2_q4.cu:465-482), the next token read from the ring by copy_embedding -- across the bins
    128 / 256, and a second sequence on the same graphs must continue the same coin stream. (The eager network is no yardstick
    for SAMPLED tokens: it is launched with the exact context length instead of the bin, another fp32 grouping in the attention,
    and a one-ulp logit moves a token whose prefix sum sits next to the coin.)"""
    prompts = ([1, 5, 9], 300), ([1, 77, 3, 12], 150)
    t = q4.Transformer(model, temperature=temperature, topp=topp, seed=4242)
    grouped = [t.generate_ids(p, n)[0].copy() for p, n in prompts]      # the RNG stream continues into the second sequence
    t.close()
    t = q4.Transformer(model, temperature=temperature, topp=topp, seed=4242)
    for (p, n), want in zip(prompts, grouped):
        t.reset(p)
        for pos in range(n):
            t.run_transformer(pos >= len(p) - 1)
            q4.synchronize()                                          # the reference order: the host reads SharedData::pos (:354)
            if pos > 0 and t.token(pos) == 2:                         # generate() stops where the ring holds EOS (:473-477), after
                break                                                 # launching that step: pos + 1 coins drawn either way
        ring = np.array([t.token(i) for i in range(n + 1)], dtype=np.int32)
        assert np.array_equal(ring[:len(want)], want), "first difference at %d" % int(np.nonzero(ring[:len(want)] != want)[0][0])
    t.close()
    assert len(set(grouped[0][4:].tolist())) > 8                      # it does sample (a constant ring would also be "equal")


def test_generate_with_temperature_runs(q4, model):
    """End to end: the CLI default (-t 0.5 -p 0.6) goes through the sampling kernel inside the captured graph."""
    t = q4.Transformer(model, temperature=0.5, topp=0.6, seed=7)
    toks, tps, timed, _ = t.generate_ids([1, 5, 9], 24)
    assert timed == 23 and len(toks) >= 24 and (toks[3:24] < t.config.vocab_size).all() and (toks[3:24] >= 0).all()
    t2 = q4.Transformer(model, temperature=0.5, topp=0.6, seed=7)
    toks2, _, _, _ = t2.generate_ids([1, 5, 9], 24)
    assert np.array_equal(toks, toks2)                                  # same seed -> same stream
    t.close()
    t2.close()
