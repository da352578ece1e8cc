import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def q4():
    """The HIP library with a stream set (GPU tests only). Fails loudly if the .so is missing."""
    from llama_cu_awq_amd import api
    import ctypes as C
    L = api.lib()
    api.DEFAULT_FUSION = L.q4_get_fusion()      # tests that change the level put this one back
    api.check(L.q4_set_device(0))
    s = C.c_void_p()
    api.check(L.q4_stream_create(C.byref(s)))
    L.q4_set_stream(s)
    yield api
    L.q4_stream_synchronize()
    L.q4_reset_graphs()
    L.q4_set_stream(None)
    L.q4_stream_destroy(s)


@pytest.fixture(autouse=True)
def _library_at_its_default_fusion_level(request):
    """Every GPU test starts at the library's default fusion level: a test that changed it and did not put it back, or a timed-out
    hand-off in an earlier test (the library then sits out some sequences at level 1), must not let later tests run at another
    level than they believe they cover."""
    if "q4" in request.fixturenames:
        api = request.getfixturevalue("q4")
        L = api.lib()
        if L.q4_get_fusion() != api.DEFAULT_FUSION:
            L.q4_set_fusion(api.DEFAULT_FUSION)
            pytest.fail("the previous test left the library at fusion level != %d" % api.DEFAULT_FUSION)
    yield


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


_OBSERVED = {}


@pytest.fixture(scope="session")
def observed():
    """Worst-case deviations the parity tests actually measured (the written tolerances are ~3x these). Dumped to
    gpurun_out/parity_observed.json at session end; the committed copy lives under profiles/."""
    import json
    yield _OBSERVED
    if _OBSERVED:
        out = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            path = os.path.join(out, "parity_observed.json")
            old = json.load(open(path)) if os.path.exists(path) else {}
            old.update(_OBSERVED)
            json.dump(old, open(path, "w"), indent=1, sort_keys=True)
        except OSError:
            pass


@pytest.fixture()
def rng():
    return np.random.default_rng(1234)


def f16_ulp_diff(a, b):
    """Distance in fp16 ulps between two float16 arrays (finite values)."""
    def key(x):
        u = np.ascontiguousarray(x, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
    return np.abs(key(a) - key(b))


def assert_close_f16(gpu, ref16, ref64=None, max_ulp=1, max_frac=0.10, what="", atol_rel=2e-5):
    """GPU fp16 vs the oracle's fp16: <= max_ulp fp16 ulps everywhere (or, for outputs that cancel to almost
    zero, an absolute difference below atol_rel * max|ref| -- the fp32 summation-order noise of a K-term sum),
    mostly identical; and, when given, vs fp64 (|err| <= 1.5 fp16 ulp of the reference + the same slack)."""
    gpu = np.asarray(gpu, dtype=np.float16)
    assert np.isfinite(gpu.astype(np.float32)).all(), what + ": non-finite output"
    ref16 = np.asarray(ref16, dtype=np.float16)
    d = f16_ulp_diff(gpu, ref16)
    atol = atol_rel * max(1.0, float(np.abs(ref16.astype(np.float64)).max()))
    d = np.where(np.abs(gpu.astype(np.float64) - ref16.astype(np.float64)) <= atol, np.minimum(d, 1), d)
    assert d.max() <= max_ulp, "%s: max fp16 ulp diff %d at %d (gpu %r ref %r)" % (
        what, d.max(), d.argmax(), gpu[d.argmax()], ref16[d.argmax()])
    assert (d > 0).mean() <= max_frac, "%s: %.3f of outputs differ from the oracle" % (what, (d > 0).mean())
    if ref64 is not None:
        err = np.abs(gpu.astype(np.float64) - ref64)
        tol = 1.5 * np.abs(ref64) * 2.0 ** -10 + 1e-4 * max(1e-6, np.abs(ref64).max()) + 6e-8   # 1.5 ulp + cancellation slack
        assert (err <= tol).all(), "%s: vs fp64 worst err %g (tol %g)" % (what, err.max(), tol[err.argmax()])
