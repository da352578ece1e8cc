"""Runs tests/prof_cases.py against the profiling build of the library (same sources, -DQ4_PROFILING) in a subprocess."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_profiling_build_cases():
    env = dict(os.environ, Q4_PROFILING_BUILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "prof_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
