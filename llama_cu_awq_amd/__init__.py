"""llama_cu_awq_amd -- MI355X-native (gfx950, HIP) llama2_q4 decode path.

The product is the C-ABI shared library `libllama2_q4.so` (include/llama2_q4.h) plus the `llama2_q4`
executable; this package is the thin ctypes mirror of the reference's host interface
(llama2_q4.cu:209-432) used by tests and bench.py. There is no CPU fallback: importing `api` raises if the
HIP library has not been built.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "api"]
