"""Cases that need the measurement knobs of the -DQ4_PROFILING build (libllama2_q4_prof.so): alternative issue orders and
slot forms of the int4 GEMV must not change results. Not collected by the default run (file name); run by
tests/test_profiling_build_gpu.py in a subprocess with Q4_PROFILING_BUILD=1, so the product library keeps exporting only
what include/llama2_q4.h declares."""
import numpy as np
import pytest

from conftest import assert_close_f16
from llama_cu_awq_amd import synth

pytestmark = pytest.mark.gpu


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
