"""Parity of the int4 GEMV family (mat_vec_kernel_int4 / qkv_matvec_kernel / ffn_matvec_silu_kernel,
gpu_kernels.h:171-275) against the CPU restatement, through the C ABI. Tolerance: <= 1 fp16 ulp vs the
oracle's lane-order fp32 result, and <= 1.5 fp16 ulp (+1e-4*max) vs an fp64 dense evaluation."""
import numpy as np
import pytest

from conftest import assert_close_f16
from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120),
          (256, 352), (352, 256), (2048, 64), (8192, 1024), (16384, 256), (32, 8),
          (22016, 8192), (28672, 8192), (32768, 512)]      # CodeLlama-34B / Llama-2-70B down projections, the K limit


def _mk(rng, K, N):
    w, z, s = synth.random_qweight(rng, K, N)
    x = rng.standard_normal(K).astype(np.float16)
    return w, z, s, x


@pytest.mark.parametrize("K,N", SHAPES)
def test_matmul_q4_plain(q4, orc, rng, K, N):
    w, z, s, x = _mk(rng, K, N)
    ref16 = orc.matmul_q4(x, w, z, s, K, N)
    ref64 = orc.matmul_q4_f64(x, w, z, s, K, N)
    dw = q4.DevQWeight(w, z, s)
    dx, dout = q4.DevBuf(x), q4.DevBuf(nbytes=N * 2)
    q4.matmul_q4(dout, dx, dw, K, N)
    q4.synchronize()
    assert_close_f16(dout.get(np.float16, N), ref16, ref64, what="plain %dx%d" % (K, N))


# (13824, 5120) and (14336, 4096): the accumulating down projections that run as strips (csrc/gemv_strip_down.h, down_strip_kernel<4, true>)
@pytest.mark.parametrize("K,N", [(4096, 4096), (11008, 4096), (352, 256), (28672, 8192), (13824, 5120), (14336, 4096)])
def test_matmul_q4_accum(q4, orc, rng, K, N):
    w, z, s, x = _mk(rng, K, N)
    old = rng.standard_normal(N).astype(np.float16)
    ref16 = orc.matmul_q4(x, w, z, s, K, N, accum_into=old)
    dw = q4.DevQWeight(w, z, s)
    dx, dout = q4.DevBuf(x), q4.DevBuf(old)
    q4.matmul_q4(dout, dx, dw, K, N, accum=True)
    q4.synchronize()
    assert_close_f16(dout.get(np.float16, N), ref16, what="accum %dx%d" % (K, N))


def test_matmul_q4_kv_addressing(q4, orc, rng):
    K, N, seq, pos, layer = 256, 256, 8, 5, 1
    w, z, s, x = _mk(rng, K, N)
    loff = layer * seq * N
    cache = np.zeros(2 * seq * N, dtype=np.float16)
    ref = cache.copy()
    orc.matmul_q4(x, w, z, s, K, N, loff=loff, pos=pos, out=ref)
    dw = q4.DevQWeight(w, z, s)
    dx, dc, dpos = q4.DevBuf(x), q4.DevBuf(cache), q4.DevBuf(np.array([pos], dtype=np.int32))
    q4.matmul_q4(dc, dx, dw, K, N, loff=loff, pPos=dpos)
    q4.synchronize()
    got = dc.get(np.float16)
    assert_close_f16(got, ref, what="kv addressing")
    assert (got[: loff + pos * N] == 0).all() and (got[loff + (pos + 1) * N:] == 0).all()


# (4096, 11008), (4096, 14336), (4096, 12552) and (4096, 12536) run as "strips" (csrc/gemv_strip.h: from 36 columns per CU on; 12552 and 12536 = ragged splits, 49 or
# 50 / 48 or 49 columns per CU), (4096, 9208) = 35 or 36 per CU just below that and (4096, 8192) stay with the wave-owned kernel
@pytest.mark.parametrize("K,N", [(4096, 11008), (5120, 13824), (256, 352), (8192, 28672), (4096, 14336), (4096, 12552), (4096, 12536), (4096, 9208), (4096, 9216), (4096, 8192)])
def test_ffn_matvec_silu(q4, orc, rng, K, N):
    g = synth.random_qweight(rng, K, N)
    u = synth.random_qweight(rng, K, N)
    x = rng.standard_normal(K).astype(np.float16)
    ref16 = orc.ffn_matvec_silu(x, g, u, K, N)
    dg, du = q4.DevQWeight(*g), q4.DevQWeight(*u)
    dx, dout = q4.DevBuf(x), q4.DevBuf(nbytes=N * 2)
    q4.ffn_matvec_silu(dout, dx, dg, du, K, N)
    q4.synchronize()
    # silu(g)*u multiplies two rounded sums: allow 2 ulp
    assert_close_f16(dout.get(np.float16, N), ref16, max_ulp=2, max_frac=0.06, what="ffn %dx%d" % (K, N))


@pytest.mark.parametrize("dim", [4096, 256, 5120])      # 5120: K in two k-slots and a shared half slot (gpu_kernels.h:213-254)
def test_qkv_matvec(q4, orc, rng, dim):
    seq, pos, layer = 6, 3, 1
    mats = [synth.random_qweight(rng, dim, dim) for _ in range(3)]
    x = rng.standard_normal(dim).astype(np.float16)
    loff = layer * seq * dim
    kc = np.zeros(2 * seq * dim, dtype=np.float16)
    vc = kc.copy()
    rq = orc.matmul_q4(x, *mats[0], dim, dim)
    rk, rv = kc.copy(), vc.copy()
    orc.matmul_q4(x, *mats[1], dim, dim, loff=loff, pos=pos, out=rk)
    orc.matmul_q4(x, *mats[2], dim, dim, loff=loff, pos=pos, out=rv)
    dm = [q4.DevQWeight(*m) for m in mats]
    dx, dq, dk, dv = q4.DevBuf(x), q4.DevBuf(nbytes=dim * 2), q4.DevBuf(kc), q4.DevBuf(vc)
    dpos = q4.DevBuf(np.array([pos], dtype=np.int32))
    q4.qkv_matvec(dq, dk, dv, dx, dm[0], dm[1], dm[2], dim, dim, loff, dpos)
    q4.synchronize()
    assert_close_f16(dq.get(np.float16, dim), rq, what="q")
    assert_close_f16(dk.get(np.float16), rk, what="k cache")
    assert_close_f16(dv.get(np.float16), rv, what="v cache")


def test_unsupported_sizes(q4, rng):
    """(n&7)||(d&7) -> "Unsupported matmul size" (llama2_q4.cu:225), reported as a status code."""
    w, z, s = synth.random_qweight(rng, 64, 16)
    dw = q4.DevQWeight(w, z, s)
    dx, dout = q4.DevBuf(nbytes=256), q4.DevBuf(nbytes=256)
    with pytest.raises(q4.Q4Error, match="Unsupported matmul size"):
        q4.matmul_q4(dout, dx, dw, 64, 12)
    with pytest.raises(q4.Q4Error, match="Unsupported matmul size"):
        q4.matmul_q4(dout, dx, dw, 60, 16)


def test_linearity_full_size(q4, rng):
    """Size-independent property at the BASELINE shape: GEMV(a*x) == a*GEMV(x) for a power of two,
    and columns are independent (permuting column blocks permutes outputs)."""
    K, N = 4096, 11008
    w, z, s = synth.random_qweight(rng, K, N)
    x = (rng.standard_normal(K) * 0.5).astype(np.float16)
    dw = q4.DevQWeight(w, z, s)
    o1, o2 = q4.DevBuf(nbytes=N * 2), q4.DevBuf(nbytes=N * 2)
    q4.matmul_q4(o1, q4.DevBuf(x), dw, K, N)
    q4.matmul_q4(o2, q4.DevBuf((x * np.float16(2)).astype(np.float16)), dw, K, N)
    q4.synchronize()
    a, b = o1.get(np.float16, N).astype(np.float32), o2.get(np.float16, N).astype(np.float32)
    # exact wherever the smaller output is a NORMAL fp16 number: doubling moves the fp32 sum's exponent, not its rounding. An output in the
    # fp16 denormal range (|y| < 2^-14: fixed spacing 2^-24) rounds its doubled sum on another grid: there one denormal step is allowed
    normal = np.abs(a) >= 2.0 ** -14
    assert np.array_equal(a[normal] * 2, b[normal])
    assert (np.abs(a[~normal] * 2 - b[~normal]) <= 2.0 ** -24).all() and (~normal).sum() < 8


class _TailQWeight:
    """A QWeight whose zeros and scales tensors END with their device allocations (sized in whole 2 MiB pages, the tensor flush against
    the end): a kernel that reads past a tensor's end leaves the allocation."""

    def __init__(self, q4, weight, zeros, scales):
        import ctypes as C
        self.bufs = [q4.DevBuf(weight)]
        ptrs = []
        for arr in (zeros, scales):
            arr = np.ascontiguousarray(arr)
            total = -(-arr.nbytes // (2 << 20)) * (2 << 20)
            buf = q4.DevBuf(nbytes=total)
            self.bufs.append(buf)
            ptr = buf.ptr + total - arr.nbytes
            q4.check(q4.lib().q4_memcpy_h2d(ptr, arr.ctypes.data, arr.nbytes))
            ptrs.append(ptr)
        self.q = q4.QWeight(self.bufs[0].ptr, ptrs[0], ptrs[1])
        self._C = C

    def ref(self):
        return self._C.byref(self.q)


@pytest.mark.parametrize("kind,K,N", [("ffn", 4096, 11008), ("ffn", 5120, 13824), ("down", 13824, 5120), ("down", 14336, 4096), ("ffn", 4096, 12552)])
def test_strips_side_data_stays_inside_tensors_that_end_with_their_allocation(q4, orc, rng, kind, K, N):
    """The strips kernels fetch their block's scales and zeros as 1 KiB LDS-DMA pieces. The last block's pieces must not reach past the
    tensors (round 4 based the descriptor at the tensor and put the block's start in soffset, which the hardware's range check does
    not cover: up to 3 KiB past the end). Results must equal the oracle with every side tensor flush against the end of its allocation."""
    x = rng.standard_normal(K).astype(np.float16)
    if kind == "ffn":
        g, u = synth.random_qweight(rng, K, N), synth.random_qweight(rng, K, N)
        ref = orc.ffn_matvec_silu(x, g, u, K, N)
        dg, du = _TailQWeight(q4, *g), _TailQWeight(q4, *u)
        dx, dout = q4.DevBuf(x), q4.DevBuf(nbytes=N * 2)
        q4.ffn_matvec_silu(dout, dx, dg, du, K, N)
        q4.synchronize()
        assert_close_f16(dout.get(np.float16, N), ref, max_ulp=2, max_frac=0.06, what="ffn %dx%d, tensors at allocation ends" % (K, N))
    else:
        w, z, s = synth.random_qweight(rng, K, N)
        old = rng.standard_normal(N).astype(np.float16)
        ref = orc.matmul_q4(x, w, z, s, K, N, accum_into=old)
        dw = _TailQWeight(q4, w, z, s)
        dx, dout = q4.DevBuf(x), q4.DevBuf(old)
        q4.matmul_q4(dout, dx, dw, K, N, accum=True)
        q4.synchronize()
        assert_close_f16(dout.get(np.float16, N), ref, what="down %dx%d, tensors at allocation ends" % (K, N))


def test_int4_gemv_has_no_systematic_error(q4, orc, rng):
    """v_dot2c_f32_f16 truncates its accumulate toward minus infinity (tools/lab/t_dot2_round.hip): left alone, every output of an int4 GEMV
    carries the same small negative error -- round 4's kernels: -2.0e-5 +- 8e-7 on this case, the restatement -4e-7 -- which the residual
    stream accumulates layer after layer (tools/error_growth.py). The kernels stage odd uint4 units negated so that even and odd lanes err
    in opposite directions (csrc/gemv_q4.h, q4_stage_sign_bits): the mean signed error over 8 x 4096 outputs of a unit-scale down
    projection must be zero within 5 standard errors, like the restatement's."""
    K, N, T = 11008, 4096, 8
    eg, er = [], []
    for t in range(T):
        w, z, sc = synth.random_qweight(rng, K, N)
        g, u = rng.standard_normal(K) * 0.9, rng.standard_normal(K) * 0.9
        x = (g / (1 + np.exp(-g)) * u).astype(np.float16)              # the down projection's input distribution: silu(g) * u
        ex = orc.matmul_q4_f64(x, w, z, sc, K, N)
        rest = orc.matmul_q4(x, w, z, sc, K, N).astype(np.float64)
        dw = q4.DevQWeight(w, z, sc)
        dx, dout = q4.DevBuf(x), q4.DevBuf(nbytes=N * 2)
        q4.matmul_q4(dout, dx, dw, K, N)
        q4.synchronize()
        eg.append(dout.get(np.float16, N).astype(np.float64) - ex)
        er.append(rest - ex)
    eg, er = np.concatenate(eg), np.concatenate(er)
    se = eg.std() / np.sqrt(len(eg))
    assert abs(eg.mean()) <= 5 * se, "systematic error %.3e (standard error %.1e; restatement %.3e)" % (eg.mean(), se, er.mean())
    assert abs(er.mean()) <= 5 * se
    assert np.sqrt(np.mean(eg ** 2)) <= 1.01 * np.sqrt(np.mean(er ** 2))
