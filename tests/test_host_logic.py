"""Host-side logic of the product through the C ABI, on the CPU: tokenizer vs goldens captured from the REFERENCE
tokenizer.h, RNG goldens (sampler.h:31-40), perplexity math (perplexity.h:3-51), CLI parser (llama2_q4.cu:624-690),
synthetic .bin layout vs the loader's size rules."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from llama_cu_awq_amd import api, synth


@pytest.fixture(scope="module")
def tok():
    t = api.Tokenizer(os.path.join(GOLDEN, "tokenizer.bin"), 32000)
    yield t
    t.close()


def test_tokenizer_encode_matches_reference_goldens(tok):
    g = json.load(open(os.path.join(GOLDEN, "tokenizer_goldens.json")))
    assert api.lib().q4_tokenizer_max_token_length(tok.h) == g["max_token_length"] == 27
    for case in g["encode"]:
        assert tok.encode(case["text"], case["bos"], case["eos"]) == case["tokens"], case["text"]


def test_tokenizer_decode_matches_reference_goldens(tok):
    g = json.load(open(os.path.join(GOLDEN, "tokenizer_goldens.json")))
    for case in g["decode"]:
        assert tok.decode(case["prev"], case["token"]).hex() == case["piece_hex"], case


def test_survey_goldens(tok):
    assert tok.encode("write an essay about GPUs", 1, 0) == [1, 2436, 385, 3686, 388, 1048, 22796, 118]
    assert tok.encode("", 1, 0) == [1]
    assert tok.encode("Once upon a time", 0, 0) == [9038, 2501, 263, 931]
    assert tok.encode("[INST] Hi [/INST]", 1, 0) == [1, 518, 25580, 96, 6324, 518, 29914, 25580, 96]


def test_missing_tokenizer_file():
    with pytest.raises(api.Q4Error):
        api.Tokenizer("/nonexistent/tokenizer.bin", 32000)


def test_rng_goldens(orc):
    g = json.load(open(os.path.join(GOLDEN, "rng_goldens.json")))
    L = api.lib()
    s = C.c_ulonglong(g["seed"])
    assert [L.random_u32(C.byref(s)) for _ in range(3)] == g["u32"]
    s = C.c_ulonglong(g["seed"])
    got = [L.random_f32(C.byref(s)) for _ in range(3)]
    assert np.allclose(got, g["f32"], rtol=0, atol=1e-8)
    # the oracle's restatement is the same stream
    s1, s2 = C.c_ulonglong(12345), C.c_ulonglong(12345)
    assert [L.random_u32(C.byref(s1)) for _ in range(100)] == [orc.lib().orc_random_u32(C.byref(s2)) for _ in range(100)]


def test_compute_perplexity_matches_the_reference_golden(orc):
    """compute_perplexity / softmax of the library and of the oracle against the output of the reference's own perplexity.h:3-51
    (tests/golden/perplexity_goldens.json, produced by make_goldens.perplexity_goldens from the reference file)."""
    import ctypes as C
    from llama_cu_awq_amd import api
    g = json.load(open(os.path.join(GOLDEN, "perplexity_goldens.json")))
    n, v = g["num_tokens"], g["vocab_size"]
    rng = np.random.default_rng(g["seed"])
    logits = (rng.standard_normal((n, v)) * g["logit_scale"]).astype(np.float32)
    tokens = rng.integers(0, v, size=n).astype(np.int32)
    L = api.lib()
    a = logits.copy()
    got = L.compute_perplexity(tokens.ctypes.data, a.ctypes.data, n, v)
    assert got == pytest.approx(g["perplexity"], rel=1e-6)
    assert np.allclose(a[0, :8], np.array(g["softmax_row0"], dtype=np.float32), rtol=1e-6, atol=0)
    assert orc.compute_perplexity(tokens, logits) == pytest.approx(g["perplexity"], rel=1e-6)


def test_compute_perplexity_matches_oracle_and_numpy(orc):
    rng = np.random.default_rng(0)
    n, v = 17, 300
    logits = (rng.standard_normal((n, v)) * 3).astype(np.float32)
    tokens = rng.integers(0, v, size=n).astype(np.int32)
    L = api.lib()
    a = logits.copy()
    got = L.compute_perplexity(tokens.ctypes.data, a.ctypes.data, n, v)
    ref = orc.compute_perplexity(tokens, logits)
    lp = logits.astype(np.float64)
    lp = lp - lp.max(axis=1, keepdims=True)
    lp = lp - np.log(np.exp(lp).sum(axis=1, keepdims=True))
    exact = float(np.exp(-lp[np.arange(n), tokens].mean()))
    assert got == pytest.approx(ref, rel=1e-6) and got == pytest.approx(exact, rel=1e-4)
    # softmax in place (perplexity.h:3-22): rows now sum to 1
    assert np.allclose(a.sum(axis=1), 1.0, atol=1e-4)


def _parse(argv):
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    out = api.CliArgs()
    rc = api.lib().q4_parse_args(len(argv), arr, C.byref(out))
    out._argv_keepalive = arr      # the parsed struct points into argv, like the reference's main()
    return rc, out


def test_cli_defaults_and_overrides():
    rc, a = _parse(["llama2_q4", "model.bin"])
    assert rc == 0 and a.checkpoint_path == b"model.bin" and a.tokenizer_path == b"tokenizer.bin"
    assert a.steps == 0 and a.temperature == pytest.approx(0.5) and a.topp == pytest.approx(0.6)   # code default 0.6 (P13)
    assert a.mode == b"generate" and a.perplexity == 0 and a.seed_from_time == 1
    rc, a = _parse(["llama2_q4", "m.bin", "-n", "256", "-i", "write an essay about GPUs", "-t", "0", "-p", "1.5", "-s", "7", "-m", "perplexity", "-q", "d.txt"])
    assert rc == 0 and a.steps == 256 and a.prompt == b"write an essay about GPUs" and a.temperature == 0.0
    assert a.topp == pytest.approx(0.9)            # out of range -> 0.9 (llama2_q4.cu:683)
    assert a.rng_seed == 7 and a.seed_from_time == 0 and a.perplexity == 1 and a.dataset_path == b"d.txt"
    rc, a = _parse(["llama2_q4", "m.bin", "-t", "-3"])
    assert rc == 0 and a.temperature == 0.0        # clamped (:682)


@pytest.mark.parametrize("argv", [["llama2_q4"], ["llama2_q4", "m.bin", "-n"], ["llama2_q4", "m.bin", "n", "3"],
                                  ["llama2_q4", "m.bin", "-nn", "3"], ["llama2_q4", "m.bin", "-x", "3"]])
def test_cli_usage_errors(argv):
    rc, _ = _parse(argv)
    assert rc == 1                                   # where the reference calls error_usage() (:640-646, :674)


def test_synth_bin_layout(tmp_path, orc):
    p = str(tmp_path / "m.bin")
    size = synth.write_model(p, "tiny", seed=3)
    assert size == os.path.getsize(p) == synth.model_bytes(synth.GEOMETRIES["tiny"])
    hdr = struct.unpack("<7if", open(p, "rb").read(32))
    assert hdr[:7] == synth.GEOMETRIES["tiny"][:7] and hdr[7] == 10000.0
    m = orc.Model(p)                                 # the oracle's loader walks the same tensor order and consumes every byte
    assert m.cfg.dim == 256 and m.cfg.hidden_dim == 352
    # wcls row 2 (EOS) is zero so greedy runs never stop early
    dim, vocab = 256, 512
    wcls = np.frombuffer(open(p, "rb").read()[32 + vocab * dim * 2: 32 + 2 * vocab * dim * 2], dtype=np.float16).reshape(vocab, dim)
    assert (wcls[2] == 0).all() and (wcls[3] != 0).any()
    m.close()
    # sizes of the real geometries
    assert synth.model_bytes(synth.GEOMETRIES["7b"]) == 3889438752
    assert synth.model_bytes(synth.GEOMETRIES["13b"]) == 7248291872


def test_product_fails_loudly_without_the_library(tmp_path, monkeypatch):
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        api.lib()
