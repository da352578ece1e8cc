"""ctypes wrapper over oracle/liboracle.so (the CPU restatement, q4_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (llama_cu_awq_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len")] + [
        ("rope_theta", C.c_float)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "q4_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def effective_cpus():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (a container that sees
    256 cores but owns 16 would otherwise oversubscribe OpenMP 16x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(min(effective_cpus(), 64))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    L = C.CDLL(build())
    L.orc_init.restype = None
    L.orc_h2f.restype = C.c_float
    L.orc_h2f.argtypes = [C.c_uint16]
    L.orc_f2h.restype = C.c_uint16
    L.orc_f2h.argtypes = [C.c_float]
    L.orc_matmul_q4.restype = None
    L.orc_matmul_q4.argtypes = [u16p, u16p, u32p, u32p, u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_matmul_q4_f64.restype = None
    L.orc_matmul_q4_f64.argtypes = [f64p, u16p, u32p, u32p, u16p, C.c_int, C.c_int]
    L.orc_ffn_matvec_silu.restype = None
    L.orc_ffn_matvec_silu.argtypes = [u16p, u16p, u32p, u32p, u16p, u32p, u32p, u16p, C.c_int, C.c_int]
    L.orc_matmul_f16.restype = None
    L.orc_matmul_f16.argtypes = [u16p, u16p, u16p, C.c_int, C.c_int, C.c_float]
    L.orc_rmsnorm.restype = None
    L.orc_rmsnorm.argtypes = [u16p, u16p, u16p, C.c_int]
    L.orc_rope.restype = None
    L.orc_rope.argtypes = [u16p, u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_attention.restype = None
    L.orc_attention.argtypes = [u16p, u16p, u16p, u16p, u16p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_attention_bin.restype = None
    L.orc_attention_bin.argtypes = [u16p, u16p, u16p, u16p, u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_seq_len_bin.restype = C.c_int
    L.orc_seq_len_bin.argtypes = [C.c_int, C.c_int]
    L.orc_argmax.restype = C.c_int
    L.orc_argmax.argtypes = [u16p, C.c_int]
    L.orc_compute_perplexity.restype = C.c_float
    L.orc_compute_perplexity.argtypes = [i32p, f32p, C.c_int, C.c_int]
    L.orc_random_u32.restype = C.c_uint
    L.orc_random_u32.argtypes = [C.POINTER(C.c_ulonglong)]
    L.orc_random_f32.restype = C.c_float
    L.orc_random_f32.argtypes = [C.POINTER(C.c_ulonglong)]
    L.orc_sample_topp.restype = C.c_int
    L.orc_sample_topp.argtypes = [u16p, C.c_int, C.c_float, C.c_float, C.c_float]
    L.orc_load.restype = C.c_void_p
    L.orc_load.argtypes = [C.c_char_p]
    L.orc_free.restype = None
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_config.restype = C.POINTER(OrcConfig)
    L.orc_config.argtypes = [C.c_void_p]
    for name in ("orc_logits", "orc_x", "orc_key_cache", "orc_value_cache"):
        getattr(L, name).restype = C.POINTER(C.c_uint16)
        getattr(L, name).argtypes = [C.c_void_p]
    L.orc_forward.restype = None
    L.orc_forward.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_set_order_variant.restype = None
    L.orc_set_order_variant.argtypes = [C.c_int]
    L.orc_set_layer_dump.restype = None
    L.orc_set_layer_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_forward_f64.restype = C.c_int
    L.orc_forward_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.orc_generate_greedy.restype = C.c_int
    L.orc_generate_greedy.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, i32p, C.c_void_p]
    L.orc_num_threads.restype = C.c_int
    L.orc_init()
    _LIB = L
    return L


def f16_bits(a):
    """numpy float16/uint16 array -> contiguous uint16 view of its bits."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        return a.view(np.uint16)
    assert a.dtype == np.uint16
    return a


def matmul_q4(x, weight, zeros, scales, inp, op, accum_into=None, loff=-1, pos=0, out=None):
    """mat_vec_kernel_int4 restatement. Returns fp16 array [op] (or writes `out` in place)."""
    L = lib()
    if out is None:
        out = np.zeros(op, dtype=np.float16) if accum_into is None else np.array(accum_into, dtype=np.float16)
    L.orc_matmul_q4(f16_bits(out), f16_bits(x), weight, zeros, f16_bits(scales), inp, op,
                    1 if accum_into is not None else 0, loff, pos)
    return out


def matmul_q4_f64(x, weight, zeros, scales, inp, op):
    out = np.zeros(op, dtype=np.float64)
    lib().orc_matmul_q4_f64(out, f16_bits(x), weight, zeros, f16_bits(scales), inp, op)
    return out


def ffn_matvec_silu(x, gate, up, inp, op):
    out = np.zeros(op, dtype=np.float16)
    lib().orc_ffn_matvec_silu(f16_bits(out), f16_bits(x), gate[0], gate[1], f16_bits(gate[2]),
                              up[0], up[1], f16_bits(up[2]), inp, op)
    return out


def matmul_f16(x, w, n, d, alpha=1.0):
    out = np.zeros(d, dtype=np.float16)
    lib().orc_matmul_f16(f16_bits(out), f16_bits(x), f16_bits(w), n, d, alpha)
    return out


def rmsnorm(x, weight):
    out = np.zeros(x.shape[0], dtype=np.float16)
    lib().orc_rmsnorm(f16_bits(out), f16_bits(x), f16_bits(weight), x.shape[0])
    return out


def rope(q, k, num_heads, num_kv_heads, head_size, pos, theta):
    """Returns rotated copies (q, k); k is this position's key row [kv_dim]."""
    q = np.array(q, dtype=np.float16)
    k = np.array(k, dtype=np.float16)
    lib().orc_rope(f16_bits(q), f16_bits(k), num_heads, num_kv_heads, head_size, pos, theta)
    return q, k


def attention(q, kc, vc, num_heads, head_size, kv_mul, pos, max_seq_len=0):
    """max_seq_len = the launch's sequence-length bin: above 8192 the reference runs softmax_kernel_no_smem (llama2_q4.cu:276-279)."""
    out = np.zeros(num_heads * head_size, dtype=np.float16)
    att = np.zeros(num_heads * (pos + 1), dtype=np.float16)
    lib().orc_attention_bin(f16_bits(out), f16_bits(q), f16_bits(kc), f16_bits(vc), f16_bits(att),
                            num_heads, head_size, kv_mul, pos, max_seq_len)
    return out, att.reshape(num_heads, pos + 1)


def argmax(x):
    return lib().orc_argmax(f16_bits(x), x.shape[0])


def compute_perplexity(tokens, logits):
    logits = np.array(logits, dtype=np.float32)  # softmax is applied in place
    tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    return lib().orc_compute_perplexity(tokens, logits, tokens.shape[0], logits.shape[1])


class Model:
    """CPU restatement of build_transformer + run_llama_network over a .bin file."""

    def __init__(self, path):
        self.L = lib()
        self.h = self.L.orc_load(path.encode())
        if not self.h:
            raise RuntimeError("orc_load failed for %s" % path)
        self.cfg = self.L.orc_config(self.h).contents

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def forward(self, token, pos):
        self.L.orc_forward(self.h, token, pos)
        return self.logits()

    def layer_dump(self, on=True):
        """tools/error_growth.py: keep the residual stream after each half layer of the last forward / forward_f64 call:
        returns (f16 [n_layers, 2, dim], f64 [n_layers, 2, dim]) arrays that the following calls fill."""
        if not on:
            self.L.orc_set_layer_dump(self.h, None, None)
            self._d16 = self._d64 = None
            return None
        self._d16 = np.zeros((self.cfg.n_layers, 2, self.cfg.dim), dtype=np.float16)
        self._d64 = np.zeros((self.cfg.n_layers, 2, self.cfg.dim), dtype=np.float64)
        self.L.orc_set_layer_dump(self.h, self._d16.ctypes.data, self._d64.ctypes.data)
        return self._d16, self._d64

    def forward_f64(self, token, pos, cap=32):
        """The same network function evaluated in double without any intermediate rounding (the yardstick for two
        valid fp16 evaluations that drift apart). Positions must be fed in order from 0; `cap` positions are kept."""
        out = np.empty(self.cfg.vocab_size, dtype=np.float64)
        rc = self.L.orc_forward_f64(self.h, int(token), int(pos), int(cap), out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc:
            raise RuntimeError("orc_forward_f64 failed: %d" % rc)
        return out

    def logits(self):
        return np.ctypeslib.as_array(self.L.orc_logits(self.h), shape=(self.cfg.vocab_size,)).view(np.float16).copy()

    def kv(self):
        kv_dim = self.cfg.dim * self.cfg.n_kv_heads // self.cfg.n_heads
        shape = (self.cfg.n_layers, self.cfg.seq_len, kv_dim)
        k = np.ctypeslib.as_array(self.L.orc_key_cache(self.h), shape=shape).view(np.float16).copy()
        v = np.ctypeslib.as_array(self.L.orc_value_cache(self.h), shape=shape).view(np.float16).copy()
        return k, v

    def generate_greedy(self, prompt, steps, want_logits=False):
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        out = np.zeros(steps + 2, dtype=np.int32)
        logits = np.zeros((steps, self.cfg.vocab_size), dtype=np.float32) if want_logits else None
        n = self.L.orc_generate_greedy(self.h, prompt, prompt.shape[0], steps, out,
                                       logits.ctypes.data_as(C.c_void_p) if want_logits else None)
        return (out[: n + 1], logits[:n] if want_logits else None)
