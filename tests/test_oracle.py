"""The CPU restatement against analytic checks and the committed fixtures (regression pins)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from llama_cu_awq_amd import synth


def test_f16_conversions_exact(orc):
    L = orc.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 7):
        v = L.orc_h2f(h)
        assert (np.isnan(v) and np.isnan(f[h])) or v == f[h]
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.standard_normal(3000).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1, 100, 30000)])
    for v in vals:
        assert L.orc_f2h(float(v)) == int(np.float32(v).astype(np.float16).view(np.uint16))
    for h in range(0, 0x7bff, 97):     # exact midpoints round to even
        mid = np.float32((float(np.uint16(h).view(np.float16)) + float(np.uint16(h + 1).view(np.float16))) / 2)
        assert L.orc_f2h(float(mid)) == int(mid.astype(np.float16).view(np.uint16))


@pytest.mark.parametrize("K,N", [(352, 64), (4096, 32), (256, 8), (11008, 16)])
def test_gemv_lane_order_vs_fp64_dense(orc, rng, K, N):
    w, z, s = synth.random_qweight(rng, K, N)
    x = rng.standard_normal(K).astype(np.float16)
    o16 = orc.matmul_q4(x, w, z, s, K, N)
    o64 = orc.matmul_q4_f64(x, w, z, s, K, N)
    dense = synth.dequant_dense(w, z, s, K, N) @ x.astype(np.float64)
    assert np.abs(o64 - dense).max() < 1e-9
    assert np.abs(o16.astype(np.float64) - o64).max() <= 2.0 ** -10 * np.abs(o64).max() + 1e-6


def test_accum_and_kv_addressing(orc, rng):
    K, N = 256, 64
    w, z, s = synth.random_qweight(rng, K, N)
    x = rng.standard_normal(K).astype(np.float16)
    base = orc.matmul_q4_f64(x, w, z, s, K, N)
    old = rng.standard_normal(N).astype(np.float16)
    acc = orc.matmul_q4(x, w, z, s, K, N, accum_into=old)
    assert np.abs(acc.astype(np.float64) - (base + old.astype(np.float64))).max() < 4e-3
    buf = np.zeros(5 * N, dtype=np.float16)
    orc.matmul_q4(x, w, z, s, K, N, loff=N, pos=2, out=buf)
    assert (buf[:3 * N] == 0).all() and (buf[4 * N:] == 0).all() and (buf[3 * N:4 * N] != 0).any()


def test_rmsnorm_rope_attention_against_numpy(orc, rng):
    x = rng.standard_normal(512).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(512)).astype(np.float16)
    xf = x.astype(np.float64)
    ref = xf / np.sqrt((xf ** 2).mean() + 1e-5) * w.astype(np.float64)
    assert np.abs(orc.rmsnorm(x, w).astype(np.float64) - ref).max() < 4e-3
    # rope: rotation preserves pair norms
    q = rng.standard_normal(4 * 64).astype(np.float16)
    k = rng.standard_normal(4 * 64).astype(np.float16)
    rq, rk = orc.rope(q, k, 4, 4, 64, 17, 10000.0)
    a, b = q.reshape(4, 2, 32).astype(np.float64), rq.reshape(4, 2, 32).astype(np.float64)
    assert np.allclose((a ** 2).sum(axis=1), (b ** 2).sum(axis=1), rtol=5e-3, atol=1e-3)
    q0, _ = orc.rope(q, k, 4, 4, 64, 0, 10000.0)
    assert np.array_equal(q0, q)                     # position 0 is the identity
    # attention vs fp64 softmax(QK^T/sqrt(d)) V
    heads, hs, pos = 4, 64, 9
    kc = rng.standard_normal((pos + 1) * heads * hs).astype(np.float16)
    vc = rng.standard_normal((pos + 1) * heads * hs).astype(np.float16)
    out, att = orc.attention(q, kc, vc, heads, hs, 1, pos)
    Q = q.reshape(heads, hs).astype(np.float64)
    K = kc.reshape(pos + 1, heads, hs).astype(np.float64)
    V = vc.reshape(pos + 1, heads, hs).astype(np.float64)
    sc = np.einsum("hd,thd->ht", Q, K) / np.sqrt(hs)
    p = np.exp(sc - sc.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    ref = np.einsum("ht,thd->hd", p, V).reshape(-1)
    assert np.abs(out.astype(np.float64) - ref).max() < 1e-2
    assert np.abs(att.astype(np.float64) - p).max() < 2e-3


def test_micro_model_fixture_regression(orc):
    """Committed fixture: the oracle must keep reproducing its own recorded logits / tokens / KV bit for bit."""
    exp = np.load(os.path.join(GOLDEN, "micro_model_expected.npz"))
    m = orc.Model(os.path.join(GOLDEN, "micro_model.bin"))
    toks, logits = m.generate_greedy(exp["prompt"], 20, want_logits=True)
    assert np.array_equal(toks, exp["tokens"])
    assert np.array_equal(logits.astype(np.float16).view(np.uint16), exp["logits"].view(np.uint16))
    k, v = m.kv()
    assert np.array_equal(k[:, :20].view(np.uint16), exp["k"].view(np.uint16))
    assert np.array_equal(v[:, :20].view(np.uint16), exp["v"].view(np.uint16))
    m.close()


def test_sampler_restatement_basic(orc, rng):
    """top-p restatement: with topp=1 the threshold is the coin itself; a peaked distribution returns its mode."""
    n = 512
    logits = rng.standard_normal(n).astype(np.float16)
    logits[77] = np.float16(30.0)
    L = orc.lib()
    assert L.orc_sample_topp(orc.f16_bits(logits.copy()), n, 1.0, 0.9, 0.5) == 77
    assert L.orc_sample_topp(orc.f16_bits(logits.copy()), n, 1.0, 1.0, 0.5) == 77


def test_unrounded_f64_forward_brackets_the_restatement(tmp_path):
    """orc_forward_f64 (double everywhere) and the reference-order fp16 restatement evaluate the same function: on a
    shallow model they agree to fp16 noise, token for token."""
    from llama_cu_awq_amd import synth
    import oracle
    p = str(tmp_path / "tiny.bin")
    synth.write_model(p, "tiny", seed=11)
    m = oracle.Model(p)
    toks = [1, 40, 22, 7, 51, 9]
    for pos, t in enumerate(toks):
        a = m.forward(t, pos).astype(np.float64)
        b = m.forward_f64(t, pos, cap=8)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 5e-3
    m.close()


@pytest.mark.parametrize("n,temperature,topp", [(1000, 1.0, 0.9), (32000, 0.5, 0.6), (32000, 1.0, 0.9), (4097, 0.7, 1.0),
                                                (32000, 0.2, 0.9), (40000, 1.0, 0.95)])
def test_topp_restatement_against_plain_numpy(n, temperature, topp):
    """The restated sampler (fp16 rounding points, fixed-order fp16 prefix scan) against the textbook float64 form:
    sort descending, cumulative sum, first index reaching coin*topp. They may only disagree inside the fp16 band of
    the prefix sum: the exact cumulative probability at the restated pick brackets the threshold to within BAND
    (measured worst 5.7e-4 = half an fp16 ulp near 1; the bound is 3x that), and wherever the threshold is further
    than BAND from both edges of the textbook pick the two agree exactly."""
    import oracle
    BAND = 2e-3
    rng = np.random.default_rng(n + int(100 * temperature))
    logits = (3.0 * rng.standard_normal(n)).astype(np.float16)
    x = (logits.astype(np.float32) / np.float32(temperature)).astype(np.float16).astype(np.float64)
    e = np.exp(x - x.max())
    p = (e / e.sum())
    order = np.lexsort((np.arange(n), -p.astype(np.float16).astype(np.float64))) if 0 < topp < 1 else np.arange(n)
    cum = np.cumsum(p[order])
    agree = 0
    coins = np.linspace(0.01, 0.99, 50)
    for coin in coins:
        thr = coin * topp if 0 < topp < 1 else coin
        tok = oracle.lib().orc_sample_topp(oracle.f16_bits(logits.copy()), n, temperature, topp, float(coin))
        k = min(int(np.searchsorted(cum, thr, side="left")), n - 1)
        pos = int(np.nonzero(order == tok)[0][0])
        lo = cum[pos - 1] if pos > 0 else 0.0
        assert lo - BAND <= thr <= cum[pos] + BAND or pos == n - 1, (coin, pos, k, lo, cum[pos], thr)
        klo = cum[k - 1] if k > 0 else 0.0
        if thr - klo > BAND and cum[k] - thr > BAND:
            assert pos == k, (coin, pos, k)
        agree += pos == k
    assert agree >= 0.7 * len(coins), agree


def test_softmax_no_smem_restatement_and_the_bin_rule(orc, rng):
    """Above 8192 positions per launch bin the reference's attention runs softmax_kernel_no_smem (llama2_q4.cu:276-279), whose
    probabilities are half(float(half(exp)) / sum) (gpu_kernels.h:432,445) instead of half(exp / sum) (:400). The restated
    probabilities must equal a plain numpy evaluation of exactly that formula on the restatement's own scores, differ from the
    <= 8192 form in some entries, and the bin must follow run_transformer's rule (:354-360)."""
    heads, hs, pos = 2, 64, 300
    q = rng.standard_normal(heads * hs).astype(np.float16)
    kc = rng.standard_normal((pos + 1) * heads * hs).astype(np.float16)
    vc = rng.standard_normal((pos + 1) * heads * hs).astype(np.float16)
    _, att_smem = orc.attention(q, kc, vc, heads, hs, 1, pos, max_seq_len=8192)
    _, att_nosmem = orc.attention(q, kc, vc, heads, hs, 1, pos, max_seq_len=16384)
    _, att_default = orc.attention(q, kc, vc, heads, hs, 1, pos)
    assert np.array_equal(att_smem, att_default)
    # recompute both from the fp16 scores: scores = half(alpha * q.k), the fp32 exp and sum as the kernels form them
    Q = q.reshape(heads, hs).astype(np.float32)
    K = kc.reshape(pos + 1, heads, hs).astype(np.float32)
    sc = (np.einsum("hd,thd->ht", Q, K).astype(np.float32) * np.float32(1.0 / np.sqrt(hs))).astype(np.float16).astype(np.float32)
    e = np.exp(sc - sc.max(axis=1, keepdims=True)).astype(np.float32)
    s = e.sum(axis=1, keepdims=True, dtype=np.float32)
    want_nosmem = (e.astype(np.float16).astype(np.float32) / s).astype(np.float16)
    want_smem = (e / s).astype(np.float16)
    from conftest import f16_ulp_diff
    assert f16_ulp_diff(att_nosmem, want_nosmem).max() <= 1 and f16_ulp_diff(att_smem, want_smem).max() <= 1   # (sum order, expf)
    assert (att_nosmem != att_smem).any()
    L = orc.lib()
    for pos_, seq, want in ((0, 2048, 128), (127, 2048, 128), (128, 2048, 256), (2047, 2048, 2048), (3000, 16384, 4096),
                            (6000, 16384, 8192), (8191, 16384, 8192), (8192, 16384, 16384), (12000, 16384, 16384), (300, 320, 512),
                            (3000, 4000, 4096), (9000, 32768, 32768)):
        assert L.orc_seq_len_bin(pos_, seq) == want, (pos_, seq)
