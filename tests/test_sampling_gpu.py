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


def test_generate_with_temperature_runs(q4, model):
    """End to end: the CLI default (-t 0.5 -p 0.6) goes through the sampling kernels outside the captured graph."""
    t = q4.Transformer(model, temperature=0.5, topp=0.6, seed=7)
    toks, tps, timed, _ = t.generate_ids([1, 5, 9], 24)
    assert timed == 23 and len(toks) >= 24 and (toks[3:24] < t.config.vocab_size).all() and (toks[3:24] >= 0).all()
    t2 = q4.Transformer(model, temperature=0.5, topp=0.6, seed=7)
    toks2, _, _, _ = t2.generate_ids([1, 5, 9], 24)
    assert np.array_equal(toks, toks2)                                  # same seed -> same stream
    t.close()
    t2.close()
