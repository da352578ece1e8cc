"""tests/consumer/hpp_consumer.cpp -- a C++ translation unit written against include/llama2_q4.hpp (the reference's
host-function names, llama2_q4.cu:209-432) -- decodes two different checkpoints on ONE Transformer object through
run_transformer's captured graphs, freeing the first with free_transformer and never calling q4_reset_graphs itself.
Tokens and the logits hash of each model must equal what the ctypes path produces for the same checkpoint."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from llama_cu_awq_amd import api, synth

pytestmark = pytest.mark.gpu


def _hash(logits):
    h = 0
    for v in logits.view(np.uint16).tolist():
        h = (h * 1000003 + v) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_consumer_decodes_frees_and_rebuilds(q4, tmp_path):
    exe = str(tmp_path / "hpp_consumer")
    libdir = os.path.dirname(api.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "consumer", "hpp_consumer.cpp"), "-o", exe,
                           "-L", libdir, "-lllama2_q4", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    paths = []
    for name, seed in (("tiny", 3), ("tiny_gqa", 4)):          # same dim, different attention geometry and weights
        p = str(tmp_path / (name + ".bin"))
        synth.write_model(p, name, seed=seed)
        paths.append(p)
    out = subprocess.check_output([exe] + paths, timeout=300).decode()
    assert "kernels ok" in out
    prompt = [1, 20, 300, 45]
    for m, p in enumerate(paths, 1):
        t = q4.Transformer(p)
        t.reset(prompt)
        for pos in range(10):
            q4.synchronize()
            t.run_transformer(pos >= len(prompt) - 1)
        q4.synchronize()
        toks = " ".join(str(int(t.token(i))) for i in range(11))
        assert ("model%d tokens %s\n" % (m, toks)) in out, (m, toks, out)
        assert ("model%d logits_hash %d pos 10" % (m, _hash(t.logits()))) in out, out
        t.close()
