"""The FFN half of a layer as ONE launch (csrc/gemv_ffn_pair.h, fusion level 4; llama2_q4.cu:326-332): rmsnorm + gate/up + SiLU, the hb vector
handed from every CU to every CU inside the launch, the down projection on weights that are already in LDS, residual add -- and, at fusion level 5,
the NEXT layer's rmsnorm + q/k/v + RoPE + KV write (llama2_q4.cu:300-317) as the launch's third phase; at the opt-in level 6 THIS layer's attention and
output projection (llama2_q4.cu:320-323) in front of it. Same arithmetic in the same order as the launches of levels 1 / 3, so everything the network
leaves behind must agree BIT FOR BIT."""
import numpy as np
import pytest

from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    d = tmp_path_factory.mktemp("ffn_pair")
    out = {}
    for name in ("ffn_pair7b", "ffn_pair_ragged", "ffn_pair_wide"):
        p = str(d / (name + ".bin"))
        synth.write_model(p, name, seed=11)
        out[name] = p
    return out


def _run(q4, path, level, graphs, steps, prompt):
    L = q4.lib()
    L.q4_set_fusion(level)
    L.q4_set_use_graphs(graphs)
    t = q4.Transformer(path)
    t.reset(prompt)
    logits, kv, hb = [], [], []
    for pos in range(steps):
        t.run_transformer(pos >= len(prompt) - 1)
        q4.synchronize()
        logits.append(t.logits().view(np.uint16).copy())
        kv.append(np.concatenate([np.concatenate(t.kv_row(l, pos)) for l in range(t.config.n_layers)]).view(np.uint16).copy())
    q4.check(L.q4_handoff_status(t.state))
    ring = [int(t.token(i)) for i in range(steps + 1)]
    t.close()
    return np.stack(logits), np.stack(kv), ring


@pytest.mark.parametrize("name,steps", [("ffn_pair7b", 140), ("ffn_pair_ragged", 40), ("ffn_pair_wide", 40)])
@pytest.mark.parametrize("graphs", [1, 0])
def test_level_4_reproduces_the_launch_sequence_bits(q4, models, name, steps, graphs):
    """Levels 4 and 3 differ only in the FFN half: logits of every position, every K / V row and the greedy token ring are identical -- across the
    bins 128 and 256 of the two-layer 7B-wide model (the epoch word is shared with the attention -> o-proj launch), eight steps per graph replay and
    eager launches, a ragged split of the column pairs over the CUs and the widest covered hidden size. No bounded wait may run out."""
    L = q4.lib()
    assert L.q4_ffn_pair_covers(4096, synth.GEOMETRIES[name][1]) == 1
    before = L.q4_handoff_timeouts()
    prompt = [1, 17, 300, 45, 9]
    try:
        a = _run(q4, models[name], 3, graphs, steps, prompt)
        b = _run(q4, models[name], 4, graphs, steps, prompt)
        assert L.q4_get_fusion() == 4 and L.q4_handoff_timeouts() == before
        c = _run(q4, models[name], 5, graphs, steps, prompt)
        assert L.q4_get_fusion() == 5 and L.q4_handoff_timeouts() == before
        d = _run(q4, models[name], 6, graphs, steps, prompt)
        assert L.q4_get_fusion() == 6 and L.q4_handoff_timeouts() == before
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
        L.q4_set_use_graphs(1)
    assert np.isfinite(a[0].view(np.float16).astype(np.float32)).all()
    assert a[2] == b[2], "greedy token rings differ"
    assert np.array_equal(a[0], b[0]), "logits differ at positions %s" % np.unique(np.argwhere(a[0] != b[0])[:, 0])[:8]
    assert np.array_equal(a[1], b[1]), "K / V rows differ at positions %s" % np.unique(np.argwhere(a[1] != b[1])[:, 0])[:8]
    # level 5: the two-layer model has ONE launch with the third phase (layer 0's FFN + layer 1's QKV); q, every K / V row and the logits are its witnesses
    assert a[2] == c[2], "greedy token rings differ (level 5)"
    assert np.array_equal(a[0], c[0]), "logits differ at positions %s (level 5)" % np.unique(np.argwhere(a[0] != c[0])[:, 0])[:8]
    assert np.array_equal(a[1], c[1]), "K / V rows differ at positions %s (level 5)" % np.unique(np.argwhere(a[1] != c[1])[:, 0])[:8]
    # level 6 (opt-in): the whole layer behind its q / k / v as one launch -- layer 0's with layer 1's QKV as its last phase, layer 1's without; the attention
    # role on sixteen waves with the P.V pass on eight, the output projection on the other half of the blocks: the same bits again, in bins 128 and 256
    assert a[2] == d[2], "greedy token rings differ (level 6)"
    assert np.array_equal(a[0], d[0]), "logits differ at positions %s (level 6)" % np.unique(np.argwhere(a[0] != d[0])[:, 0])[:8]
    assert np.array_equal(a[1], d[1]), "K / V rows differ at positions %s (level 6)" % np.unique(np.argwhere(a[1] != d[1])[:, 0])[:8]


def test_level_4_against_the_restatement(q4, orc, models):
    """... and against the CPU restatement of run_llama_network (oracle/), the model's bound of tests/test_forward_gpu.py."""
    L = q4.lib()
    try:
        L.q4_set_fusion(5)
        t = q4.Transformer(models["ffn_pair7b"])
        m = orc.Model(models["ffn_pair7b"])
        prompt = [1, 17, 300, 45, 9]
        t.reset(prompt)
        toks = list(prompt)
        for pos in range(10):
            gen = pos >= len(prompt) - 1
            t.run_transformer(gen)
            q4.synchronize()
            ref = m.forward(toks[pos], pos).astype(np.float64)
            got = t.logits().astype(np.float64)
            assert (np.abs(got - ref) <= 5e-3 * np.maximum(1.0, np.abs(ref))).all(), (pos, float(np.abs(got - ref).max()))
            if gen:
                toks.append(t.token(pos + 1))
        t.close()
        m.close()
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)


def test_shapes_the_launch_does_not_cover_run_the_launch_sequence(q4, tmp_path):
    """Level 4 on a model whose FFN the launch does not cover (dim 2560) is level 3."""
    L = q4.lib()
    assert L.q4_ffn_pair_covers(2560, 3584) == 0 and L.q4_ffn_pair_covers(5120, 13824) == 0 and L.q4_ffn_pair_covers(4096, 14336) == 0
    p = str(tmp_path / "head128.bin")
    synth.write_model(p, "head128", seed=7)
    try:
        a = _run(q4, p, 3, 1, 20, [1, 5, 9])
        b = _run(q4, p, 5, 1, 20, [1, 5, 9])
    finally:
        L.q4_set_fusion(q4.DEFAULT_FUSION)
    assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
