"""SURVEY 8f-1 closed on the GPU: synthetic AWQ dump (old and new tensor formats, MHA and GQA) -> this build's
weight_packer (weight_packer.cpp:233-297) -> q4_build_transformer (checkpoint_init_weights llama2_q4.cu:172-202) ->
decode steps, against the CPU restatement loading the SAME packed file; then the llama2_q4 executable on it."""
import os
import re
import subprocess

import numpy as np
import pytest

import packer_util
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
PACKER = os.path.join(ROOT, "llama_cu_awq_amd", "bin", "weight_packer")
EXE = os.path.join(ROOT, "llama_cu_awq_amd", "bin", "llama2_q4")
TOK = os.path.join(GOLDEN, "tokenizer.bin")


def _pack(tmp_path, fmt, seed, cfg):
    cfg = packer_util.write_awq_dump(str(tmp_path), old_format=bool(fmt), seed=seed, cfg=cfg)
    dst = str(tmp_path / "packed.bin")
    subprocess.check_call([PACKER, str(tmp_path / "config.json"), str(tmp_path / "dump"), dst, str(fmt)], stdout=subprocess.DEVNULL)
    return dst, cfg


@pytest.mark.parametrize("fmt,seed,gqa", [(1, 11, False), (0, 12, False), (1, 13, True), (0, 14, True)])
def test_packed_checkpoint_decodes_like_the_restatement(q4, orc, tmp_path, fmt, seed, gqa, observed):
    cfg = dict(packer_util.CFG, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8,
               num_key_value_heads=2 if gqa else 8, vocab_size=512, max_position_embeddings=64)
    if gqa:
        cfg["rope_theta"] = 1000000.0
    path, cfg = _pack(tmp_path, fmt, seed, cfg)
    t = q4.Transformer(path)
    m = orc.Model(path)
    c = t.config
    assert (c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.vocab_size, c.seq_len) == (512, 1408, 2, 8, 2 if gqa else 8, 512, 64)
    assert c.rope_theta == pytest.approx(cfg["rope_theta"])
    prompt = [1, 17, 300, 45]
    t.reset(prompt)
    toks = list(prompt)
    worst = 0.0
    for pos in range(8):
        gen = pos >= len(prompt) - 1
        t.run_transformer(gen)
        q4.synchronize()
        ref = m.forward(toks[pos], pos).astype(np.float64)
        got = t.logits().astype(np.float64)
        # the dump's norm weights, embedding AND lm_head are N(0,1): the residual reaches O(50), one fp16 ulp of it (0.03)
        # times an O(1) classifier weight moves any logit by that much, whatever its own size -- the error scales with
        # the LARGEST logit. Measured 1.2e-3 of max|logit| (2 fp16 ulps of it); bound = 3x
        scale = float(np.abs(ref).max())
        worst = max(worst, float(np.abs(got - ref).max()) / scale)
        assert np.abs(got - ref).max() <= 3.6e-3 * scale, (pos, np.abs(got - ref).max(), scale)
        if gen:
            toks.append(int(t.token(pos + 1)))
            top2 = np.sort(ref)[-2:]
            if top2[1] - top2[0] > 4e-3 * max(1.0, abs(top2[1])):
                assert toks[-1] == int(np.argmax(ref))
    observed["packed_checkpoint_fmt%d_%s" % (fmt, "gqa" if gqa else "mha")] = {"logits_max_err_over_max_logit": worst}
    t.close()
    m.close()


def test_cli_runs_a_packed_checkpoint(q4, orc, tmp_path):
    cfg = dict(packer_util.CFG, hidden_size=256, intermediate_size=352, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=4, vocab_size=32000, max_position_embeddings=128)
    path, cfg = _pack(tmp_path, 1, 21, cfg)
    r = subprocess.run([EXE, path, "-n", "24", "-i", "write an essay about GPUs", "-t", "0", "-z", TOK], capture_output=True, text=True,
                       timeout=300, errors="replace")
    assert r.returncode == 0, r.stderr
    assert "dim: 256 \nhidden_dim: 352" in r.stdout and "Loading Weights... done!" in r.stdout
    m = re.search(r"achieved tok/s: ([0-9.]+)\. Tokens: (\d+)", r.stdout)
    assert m and int(m.group(2)) == 23
