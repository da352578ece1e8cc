"""Parity of the fp16 companion kernels (rmsnorm, fp16 GEMV, RoPE, attention, embedding, argmax, fp16->fp32)
against the CPU restatement, through the C ABI."""
import numpy as np
import pytest

from conftest import assert_close_f16, f16_ulp_diff

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [4096, 5120, 256, 8])
def test_rmsnorm(q4, orc, rng, size):
    x = (rng.standard_normal(size) * 3).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(size)).astype(np.float16)
    ref = orc.rmsnorm(x, w)
    dx, dw, do = q4.DevBuf(x), q4.DevBuf(w), q4.DevBuf(nbytes=size * 2)
    q4.rmsnorm(do, dx, dw, size)
    q4.synchronize()
    assert_close_f16(do.get(np.float16, size), ref, what="rmsnorm")
    q4.rmsnorm(dx, dx, dw, size)        # in place, as the final norm (llama2_q4.cu:336)
    q4.synchronize()
    assert_close_f16(dx.get(np.float16, size), ref, what="rmsnorm in place")


# (4096, 32000), (5120, 32008) and (4096, 16384) run as strips (csrc/gemv_strip_cls.h: n = 4096 / 5120, at least 64 rows per CU; 32008 = a ragged split
# of rows over the CUs), the others as gemv_f16_kernel
@pytest.mark.parametrize("n,d", [(4096, 32000), (5120, 1000), (256, 512), (2048, 64), (5120, 32008), (4096, 16384), (4096, 16376)])
def test_matmul_f16(q4, orc, rng, n, d):
    w = (rng.standard_normal(n * d) * 0.02).astype(np.float16)
    x = rng.standard_normal(n).astype(np.float16)
    ref = orc.matmul_f16(x, w, n, d)
    ref64 = w.reshape(d, n).astype(np.float64) @ x.astype(np.float64)
    dw, dx, do = q4.DevBuf(w), q4.DevBuf(x), q4.DevBuf(nbytes=d * 2)
    q4.matmul(do, dx, dw, n, d)
    q4.synchronize()
    assert_close_f16(do.get(np.float16, d), ref, ref64, what="fp16 gemv %dx%d" % (n, d))


@pytest.mark.parametrize("heads,kv_heads,hs,pos", [(32, 32, 128, 0), (32, 32, 128, 255), (8, 2, 64, 17), (4, 4, 64, 2047)])
def test_rope(q4, orc, rng, heads, kv_heads, hs, pos):
    q = rng.standard_normal(heads * hs).astype(np.float16)
    seq = pos + 2
    kv_dim = kv_heads * hs
    kc = rng.standard_normal(2 * seq * kv_dim).astype(np.float16)
    loff = seq * kv_dim
    krow = kc[loff + pos * kv_dim: loff + (pos + 1) * kv_dim]
    rq, rk = orc.rope(q, krow, heads, kv_heads, hs, pos, 10000.0)
    dq, dk, dpos = q4.DevBuf(q), q4.DevBuf(kc), q4.DevBuf(np.array([pos], dtype=np.int32))
    q4.RoPERotation(dq, dk, heads, kv_heads, hs, dpos, loff, 10000.0)
    q4.synchronize()
    gk = dk.get(np.float16)
    # device sinf/cosf/powf vs glibc: a few fp32 ulp on angles up to 2047 rad -> at most 1 fp16 ulp after rounding
    assert_close_f16(dq.get(np.float16), rq, max_frac=0.05, what="rope q")
    assert_close_f16(gk[loff + pos * kv_dim: loff + (pos + 1) * kv_dim], rk, max_frac=0.05, what="rope k")
    untouched = np.ones(gk.shape[0], dtype=bool)
    untouched[loff + pos * kv_dim: loff + (pos + 1) * kv_dim] = False
    assert np.array_equal(gk[untouched], kc[untouched])


def _attention_close(got, ref, frac_gt1=0.03):
    """fp16 ulps, not a flat tolerance: expf ulps and the summation order over fp16-rounded probabilities move an output
    by at most a few ulps, except where the weighted sum cancels to ~0 (there: absolute, 3x the measured 6.1e-5).
    Measured (tools/measure_tolerances.py): <= 3 ulps up to 256 positions, 1.1 % of outputs off by more than one ulp at
    16 K positions, worst |err| 6.1e-5 on outputs of <= 0.4."""
    d = f16_ulp_diff(got, ref)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert ((d <= 4) | (err <= 2e-4)).all(), (int(d.max()), float(err.max()))
    assert (d > 1).mean() <= frac_gt1, float((d > 1).mean())


@pytest.mark.parametrize("heads,kv_mul,hs,pos,seq", [(32, 1, 128, 0, 128), (32, 1, 128, 255, 256), (32, 1, 128, 300, 512),
                                                      (8, 4, 64, 40, 128), (4, 1, 64, 63, 64), (4, 2, 256, 9, 128),
                                                      (8, 1, 32, 20, 128), (32, 1, 128, 2047, 2048),
                                                      (32, 1, 128, 127, 128), (8, 2, 128, 77, 128), (32, 1, 128, 128, 256),
                                                      # head sizes that are no power of two (the reference's kernels take any: :267-284)
                                                      (8, 1, 96, 50, 128), (8, 2, 80, 300, 512), (4, 1, 160, 700, 1024), (6, 3, 24, 17, 64),
                                                      (4, 1, 200, 1500, 2048)])
def test_attention(q4, orc, rng, heads, kv_mul, hs, pos, seq):
    dim = heads * hs
    kv_dim = dim // kv_mul
    q = rng.standard_normal(dim).astype(np.float16)
    kc = rng.standard_normal(seq * kv_dim).astype(np.float16)
    vc = rng.standard_normal(seq * kv_dim).astype(np.float16)
    ref, _ = orc.attention(q, kc, vc, heads, hs, kv_mul, pos)
    dq, dk, dv, do = q4.DevBuf(q), q4.DevBuf(kc), q4.DevBuf(vc), q4.DevBuf(nbytes=dim * 2)
    dpos = q4.DevBuf(np.array([pos], dtype=np.int32))
    q4.MultiHeadAttention(do, dq, dk, dv, None, heads, hs, kv_mul, seq, dpos)
    q4.synchronize()
    got = do.get(np.float16, dim)
    _attention_close(got, ref)


@pytest.mark.parametrize("hs,kv_mul", [(128, 1), (64, 1), (64, 4), (256, 2)])
@pytest.mark.parametrize("pos,seq", [(2047, 2048), (1000, 2048), (100, 1024), (1023, 1024), (4000, 4096)])
def test_attention_split_context(q4, orc, rng, pos, seq, hs, kv_mul):
    """Long-context path: one block per (head, 256-position chunk) + combine, scratch = the `att` buffer; head sizes 64,
    128 and 256, with grouped-query attention."""
    heads = 4096 // hs if hs != 256 else 8
    dim = heads * hs
    q = rng.standard_normal(dim).astype(np.float16)
    kc = rng.standard_normal(seq * dim // kv_mul).astype(np.float16)
    vc = rng.standard_normal(seq * dim // kv_mul).astype(np.float16)
    ref, _ = orc.attention(q, kc, vc, heads, hs, kv_mul, pos)
    dq, dk, dv, do = q4.DevBuf(q), q4.DevBuf(kc), q4.DevBuf(vc), q4.DevBuf(nbytes=dim * 2)
    att = q4.DevBuf(nbytes=heads * max(seq, dim) * 2 * 2)
    dpos = q4.DevBuf(np.array([pos], dtype=np.int32))
    q4.check(q4.lib().q4_multi_head_attention(do.ptr, dq.ptr, dk.ptr, dv.ptr, att.ptr, heads, hs, kv_mul, seq * 2 if seq < 4096 else seq, dpos.ptr))
    q4.synchronize()
    got = do.get(np.float16, dim)
    # the split form keeps the probabilities in fp32 where the reference's `att` buffer rounds them to fp16 (:400): up to
    # 2^-11 relative per term, which moves ~13 % of these tiny (|out| ~ 0.03) outputs by 2 fp16 ulps (measured); the
    # absolute bound is the same as for the one-block kernel
    _attention_close(got, ref, frac_gt1=0.25)


def test_copy_embedding_and_convert(q4, rng):
    size, vocab = 4096, 64
    table = rng.standard_normal(vocab * size).astype(np.float16)
    tokens = np.array([3, 7, 42, 1], dtype=np.int32)
    dt, dx = q4.DevBuf(table), q4.DevBuf(nbytes=size * 2)
    dtok, dpos = q4.DevBuf(tokens), q4.DevBuf(np.array([2], dtype=np.int32))
    q4.check(q4.lib().q4_copy_embedding(dx.ptr, dt.ptr, size, dtok.ptr, dpos.ptr))
    q4.synchronize()
    assert np.array_equal(dx.get(np.float16, size), table[42 * size: 43 * size])
    df = q4.DevBuf(nbytes=size * 4)
    q4.check(q4.lib().q4_convert_fp16_to_fp32(df.ptr, dx.ptr, size))
    q4.synchronize()
    assert np.array_equal(df.get(np.float32, size), table[42 * size: 43 * size].astype(np.float32))


@pytest.mark.parametrize("size", [32000, 512, 1000])
def test_argmax(q4, orc, rng, size):
    x = rng.standard_normal(size).astype(np.float16)
    x[[5, size - 3]] = x.max() + np.float16(1)     # a tie: lowest index must win
    dx = q4.DevBuf(x)
    ring = q4.DevBuf(np.zeros(16, dtype=np.int32))
    hpos, dpos = q4.DevBuf(np.array([4], dtype=np.int32)), q4.DevBuf(np.array([4], dtype=np.int32))   # host + device copies of pos
    q4.check(q4.lib().q4_argmax(dx.ptr, size, ring.ptr, hpos.ptr, dpos.ptr, 1))
    q4.synchronize()
    assert orc.argmax(x) == 5
    r = ring.get(np.int32)
    assert r[5] == 5 and hpos.get(np.int32)[0] == 5 and dpos.get(np.int32)[0] == 5
    q4.check(q4.lib().q4_argmax(dx.ptr, size, ring.ptr, hpos.ptr, dpos.ptr, 0))   # prompt phase: only advance pos
    q4.synchronize()
    assert ring.get(np.int32)[6] == 0 and hpos.get(np.int32)[0] == 6


@pytest.mark.parametrize("with_scratch", [False, True])
def test_attention_16k_context(q4, orc, rng, with_scratch):
    """contexts past the reference's 8192-position shared-memory softmax (softmax_kernel_no_smem,
    gpu_kernels.h:403-446): 16384-position cache, GQA (kv_mul 2), both the one-block-per-head kernel (scores in LDS)
    and the split-context kernels (partials in the `att` scratch)."""
    heads, kv_mul, hs, pos, seq = 8, 2, 128, 16000, 16384
    dim, kv_dim = heads * hs, heads * hs // kv_mul
    q = rng.standard_normal(dim).astype(np.float16)
    kc = (0.5 * rng.standard_normal(seq * kv_dim)).astype(np.float16)
    vc = rng.standard_normal(seq * kv_dim).astype(np.float16)
    ref, _ = orc.attention(q, kc, vc, heads, hs, kv_mul, pos, max_seq_len=seq)     # > 8192: the restated softmax_kernel_no_smem
    dq, dk, dv, do = q4.DevBuf(q), q4.DevBuf(kc), q4.DevBuf(vc), q4.DevBuf(nbytes=dim * 2)
    dpos = q4.DevBuf(np.array([pos], dtype=np.int32))
    att = q4.DevBuf(nbytes=heads * max(seq, dim) * 2 * 2) if with_scratch else None
    q4.check(q4.lib().q4_multi_head_attention(do.ptr, dq.ptr, dk.ptr, dv.ptr, att.ptr if att else None, heads, hs, kv_mul, seq, dpos.ptr))
    q4.synchronize()
    got = do.get(np.float16, dim)
    _attention_close(got, ref)
