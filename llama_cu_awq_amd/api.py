"""ctypes mirror of include/llama2_q4.h (same names / argument meaning as the reference's host functions).

Fails loudly when libllama2_q4.so is missing -- there is no CPU or PyTorch fallback on this path.
Device buffers are plain HIP allocations owned by `DevBuf`; arrays cross the boundary as numpy.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libllama2_q4.so")
# the -DQ4_PROFILING build of the same sources (ablations, launch-skip mask, tuning setters): tools/ and
# tests/prof_cases.py select it with Q4_PROFILING_BUILD=1 or use_profiling_build() before the first lib() call
PROF_LIB_PATH = os.path.join(_HERE, "libllama2_q4_prof.so")

MAX_SEQ_LEN = 128 * 1024


class Config(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len")] + [
        ("rope_theta", C.c_float)]


class QWeight(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("zeros", C.c_void_p), ("scales", C.c_void_p)]


class PerLayerWeight(C.Structure):
    _fields_ = [("rms_att_weight", C.c_void_p), ("rms_ffn_weight", C.c_void_p)] + [
        (n, QWeight) for n in ("wq_q", "wq_k", "wq_v", "wq_o", "wq_gate", "wq_up", "wq_down")]


class TransformerWeights(C.Structure):
    _fields_ = [("token_embedding_table", C.c_void_p), ("wcls", C.c_void_p), ("rms_final_weight", C.c_void_p),
                ("layers", C.POINTER(PerLayerWeight)), ("num_layers", C.c_int)]


class RunState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "xb", "hb", "q", "att", "logits", "key_cache", "value_cache", "pos",
                                          "shared_data", "logits_array")]


class CliArgs(C.Structure):
    _fields_ = [("checkpoint_path", C.c_char_p), ("tokenizer_path", C.c_char_p), ("dataset_path", C.c_char_p),
                ("steps", C.c_int), ("prompt", C.c_char_p), ("perplexity", C.c_int), ("temperature", C.c_float),
                ("topp", C.c_float), ("rng_seed", C.c_ulonglong), ("mode", C.c_char_p), ("system_prompt", C.c_char_p),
                ("seed_from_time", C.c_int)]


# every symbol include/llama2_q4.h declares (tests/test_abi.py checks the .so exports all of them)
SYMBOLS = [
    "q4_status_string", "q4_last_error", "q4_set_device", "q4_stream_create", "q4_stream_create_masked", "q4_stream_destroy", "q4_set_stream",
    "q4_get_stream", "q4_stream_synchronize", "q4_device_synchronize", "q4_malloc", "q4_free", "q4_memcpy_h2d",
    "q4_memcpy_d2h", "q4_memset", "q4_rmsnorm", "q4_matmul_f16", "q4_matmul_q4", "q4_qkv_matvec", "q4_ffn_matvec_silu",
    "q4_rope_rotation", "q4_multi_head_attention", "q4_copy_embedding", "q4_convert_fp16_to_fp32", "q4_argmax",
    "q4_run_llama_network", "q4_run_transformer", "q4_run_transformer_at", "q4_wait_pos", "q4_steps_that_fit", "q4_run_transformer_steps", "q4_set_fusion", "q4_get_fusion", "q4_ffn_pair_covers", "q4_set_use_graphs", "q4_reset_graphs", "q4_graph_captures",
    "build_sampler", "destroy_sampler", "random_u32", "random_f32", "q4_sample", "q4_build_transformer",
    "q4_free_transformer", "q4_set_quiet", "q4_transformer_new", "q4_transformer_delete", "q4_transformer_config",
    "q4_transformer_state", "q4_transformer_weights", "q4_sampler_new", "q4_sampler_delete", "q4_reset_sequence",
    "q4_shared_pos", "q4_handoff_status", "q4_handoff_timeouts", "q4_kv_stream_price", "q4_shared_token", "q4_get_logits", "q4_get_kv_row", "q4_get_logits_array", "q4_generate",
    "q4_generate_ids", "q4_chat", "q4_softmax_f32", "compute_perplexity", "q4_get_dataset_perplexity",
    "q4_parse_dataset_and_compute_perplexity", "q4_perplexity_ids", "q4_tokenizer_new", "q4_tokenizer_delete",
    "q4_tokenizer_encode", "q4_tokenizer_decode", "q4_tokenizer_max_token_length", "q4_main", "q4_parse_args",
    "q4_bench_kernel", "q4_bench_kernel_graph", "q4_bench_in_network", "q4_device_info",
]

_lib = None
_use_prof = os.environ.get("Q4_PROFILING_BUILD", "") == "1"


def use_profiling_build():
    global _use_prof
    assert _lib is None, "select the profiling build before the library is loaded"
    _use_prof = True


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = PROF_LIB_PATH if _use_prof else LIB_PATH
    # tools/ab.py compares builds of the library inside one gpurun call (clocks differ from box to box by several per cent)
    path = os.environ.get("Q4_LIB_OVERRIDE") or path
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C llama_cu_awq_amd/csrc`. There is no CPU fallback for this path." % (os.path.basename(path), path))
    L = C.CDLL(path)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.q4_status_string.restype = C.c_char_p
    L.q4_status_string.argtypes = [i]
    L.q4_last_error.restype = C.c_char_p
    L.q4_stream_create.argtypes = [C.POINTER(vp)]
    if hasattr(L, "q4_stream_create_masked"):      # (older builds under tools/ab.py do not have it)
        L.q4_stream_create_masked.argtypes = [C.POINTER(vp), i]
    L.q4_stream_destroy.argtypes = [vp]
    L.q4_set_stream.argtypes = [vp]
    L.q4_set_stream.restype = None
    L.q4_get_stream.restype = vp
    L.q4_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.q4_free.argtypes = [vp]
    L.q4_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.q4_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.q4_memset.argtypes = [vp, i, C.c_size_t]
    L.q4_rmsnorm.argtypes = [vp, vp, vp, i]
    L.q4_matmul_f16.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, f]
    L.q4_matmul_q4.argtypes = [vp, vp, C.POINTER(QWeight), i, i, i, i, vp]
    L.q4_qkv_matvec.argtypes = [vp, vp, vp, vp, C.POINTER(QWeight), C.POINTER(QWeight), C.POINTER(QWeight), i, i, i, vp]
    L.q4_ffn_matvec_silu.argtypes = [vp, vp, C.POINTER(QWeight), C.POINTER(QWeight), i, i]
    L.q4_rope_rotation.argtypes = [vp, vp, i, i, i, vp, i, f]
    L.q4_multi_head_attention.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    L.q4_copy_embedding.argtypes = [vp, vp, i, vp, vp]
    L.q4_convert_fp16_to_fp32.argtypes = [vp, vp, i]
    L.q4_argmax.argtypes = [vp, i, vp, vp, vp, i]
    L.q4_run_llama_network.argtypes = [vp, C.POINTER(Config), C.POINTER(RunState), C.POINTER(TransformerWeights), i]
    L.q4_run_transformer.argtypes = [i, C.POINTER(Config), C.POINTER(RunState), C.POINTER(TransformerWeights), i, vp]
    L.q4_set_fusion.argtypes = [i]
    L.q4_set_fusion.restype = None
    if hasattr(L, "q4_kv_stream_price"):
        L.q4_kv_stream_price.restype = C.c_double
        L.q4_kv_stream_price.argtypes = [vp]
    if hasattr(L, "q4_ffn_pair_covers"):           # (older builds under tools/ab.py do not have it)
        L.q4_ffn_pair_covers.argtypes = [i, i]
    L.q4_set_use_graphs.argtypes = [i]
    L.q4_set_use_graphs.restype = None
    L.q4_reset_graphs.restype = None
    L.q4_set_quiet.argtypes = [i]
    L.q4_set_quiet.restype = None
    L.random_u32.argtypes = [C.POINTER(C.c_ulonglong)]
    L.random_u32.restype = C.c_uint
    L.random_f32.argtypes = [C.POINTER(C.c_ulonglong)]
    L.random_f32.restype = f
    L.q4_transformer_new.argtypes = [C.c_char_p, i, C.POINTER(i)]
    L.q4_transformer_new.restype = vp
    L.q4_transformer_delete.argtypes = [vp]
    L.q4_transformer_delete.restype = None
    L.q4_transformer_config.argtypes = [vp]
    L.q4_transformer_config.restype = C.POINTER(Config)
    L.q4_transformer_state.argtypes = [vp]
    L.q4_transformer_state.restype = C.POINTER(RunState)
    L.q4_transformer_weights.argtypes = [vp]
    L.q4_transformer_weights.restype = C.POINTER(TransformerWeights)
    L.q4_sampler_new.argtypes = [i, f, f, C.c_ulonglong]
    L.q4_sampler_new.restype = vp
    L.q4_sampler_delete.argtypes = [vp]
    L.q4_sampler_delete.restype = None
    L.q4_sample.argtypes = [vp, C.POINTER(RunState), i]
    L.q4_reset_sequence.argtypes = [C.POINTER(RunState), vp, i]
    L.q4_shared_pos.argtypes = [C.POINTER(RunState)]
    L.q4_shared_token.argtypes = [C.POINTER(RunState), i]
    L.q4_handoff_status.argtypes = [C.POINTER(RunState)]
    L.q4_get_logits.argtypes = [vp, vp]
    L.q4_get_kv_row.argtypes = [vp, i, i, vp, vp]
    L.q4_get_logits_array.argtypes = [vp, i, vp]
    L.q4_generate_ids.argtypes = [vp, vp, vp, i, i, vp, C.POINTER(i), C.POINTER(C.c_double)]
    L.q4_generate_ids.restype = C.c_double
    L.q4_generate.argtypes = [vp, vp, vp, C.c_char_p, i, C.POINTER(i), C.POINTER(C.c_double)]
    L.q4_generate.restype = C.c_double
    L.q4_perplexity_ids.argtypes = [vp, vp, vp, i]
    L.q4_perplexity_ids.restype = f
    L.q4_softmax_f32.argtypes = [vp, i]
    L.q4_softmax_f32.restype = None
    L.compute_perplexity.argtypes = [vp, vp, i, i]
    L.compute_perplexity.restype = f
    L.q4_tokenizer_new.argtypes = [C.c_char_p, i]
    L.q4_tokenizer_new.restype = vp
    L.q4_tokenizer_delete.argtypes = [vp]
    L.q4_tokenizer_delete.restype = None
    L.q4_tokenizer_encode.argtypes = [vp, C.c_char_p, i, i, vp, C.POINTER(i)]
    L.q4_tokenizer_decode.argtypes = [vp, i, i]
    L.q4_tokenizer_decode.restype = C.c_char_p
    L.q4_tokenizer_max_token_length.argtypes = [vp]
    L.q4_parse_args.argtypes = [i, C.POINTER(C.c_char_p), C.POINTER(CliArgs)]
    L.q4_main.argtypes = [i, C.POINTER(C.c_char_p)]
    L.q4_bench_kernel.argtypes = [i, C.POINTER(Config), C.POINTER(RunState), C.POINTER(TransformerWeights), i,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.q4_bench_kernel.restype = C.c_double
    L.q4_bench_kernel_graph.argtypes = [i, C.POINTER(Config), C.POINTER(RunState), C.POINTER(TransformerWeights), i, i]
    L.q4_bench_kernel_graph.restype = C.c_double
    L.q4_bench_in_network.argtypes = [i, i, C.POINTER(Config), C.POINTER(RunState), C.POINTER(TransformerWeights), i,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i)]
    L.q4_bench_in_network.restype = C.c_double
    L.q4_device_info.argtypes = [C.c_char_p, i, C.POINTER(i), C.POINTER(C.c_size_t)]
    if _use_prof:
        for name, at in (("q4_set_gemv_tune", [i, i, i]), ("q4_set_gemv_early", [i, i]), ("q4_set_half_tail", [i]),
                         ("q4_set_ksplit", [i]), ("q4_set_ablate", [i]), ("q4_set_skip_mask", [i]),
                         ("q4_set_attention_split", [i, i]), ("q4_set_debug_buffer", [vp])):
            getattr(L, name).argtypes = at
            getattr(L, name).restype = None
    _lib = L
    return L


class Q4Error(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        L = lib()
        raise Q4Error("%s (status %d) %s" % (L.q4_status_string(rc).decode(), rc, L.q4_last_error().decode()))


class DevBuf:
    """A HIP device allocation holding a numpy array's bytes."""

    def __init__(self, arr=None, nbytes=None):
        L = lib()
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes = arr.nbytes
        self.nbytes = max(int(nbytes), 16)
        p = C.c_void_p()
        check(L.q4_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value
        if arr is not None:
            check(L.q4_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes))
        else:
            check(L.q4_memset(self.ptr, 0, self.nbytes))
            check(L.q4_stream_synchronize())

    def get(self, dtype, count=None):
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(n, dtype=dtype)
        check(lib().q4_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().q4_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes))

    def free(self):
        if self.ptr:
            lib().q4_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DevQWeight:
    """QWeight (common.h:20-24) on the device, from (weight u32, zeros u32, scales f16) numpy arrays."""

    def __init__(self, weight, zeros, scales):
        self.bufs = [DevBuf(weight), DevBuf(zeros), DevBuf(scales)]
        self.q = QWeight(self.bufs[0].ptr, self.bufs[1].ptr, self.bufs[2].ptr)

    def ref(self):
        return C.byref(self.q)


# ---- the reference's host functions, same names and argument order (llama2_q4.cu:209-284) ----------------
def rmsnorm(o, x, weight, size):
    check(lib().q4_rmsnorm(o.ptr, x.ptr, weight.ptr, size))


def matmul(xout, x, w, n, d, batch=1, x_stride=0, w_stride=0, op_stride=0, w_row_stride=-1, alpha=1.0):
    check(lib().q4_matmul_f16(xout.ptr, x.ptr, w.ptr, n, d, batch, x_stride, w_stride, op_stride, w_row_stride, alpha))


def matmul_q4(xout, x, w, inpSize, opSize, accum=False, loff=-1, pPos=None):
    check(lib().q4_matmul_q4(xout.ptr, x.ptr, w.ref(), inpSize, opSize, int(accum), loff, pPos.ptr if pPos else None))


def qkv_matvec(q, key_cache, value_cache, x, qw, kw, vw, inpSize, opSize, loff, pPos):
    check(lib().q4_qkv_matvec(q.ptr, key_cache.ptr, value_cache.ptr, x.ptr, qw.ref(), kw.ref(), vw.ref(), inpSize, opSize,
                              loff, pPos.ptr))


def ffn_matvec_silu(xout, x, gate_w, up_w, inpSize, opSize):
    check(lib().q4_ffn_matvec_silu(xout.ptr, x.ptr, gate_w.ref(), up_w.ref(), inpSize, opSize))


def RoPERotation(q, k, num_heads, num_kv_heads, head_size, pPos, loff, rope_theta):
    check(lib().q4_rope_rotation(q.ptr, k.ptr, num_heads, num_kv_heads, head_size, pPos.ptr, loff, rope_theta))


def MultiHeadAttention(output, q, key_cache, value_cache, att, num_heads, head_size, kv_mul, max_seq_len, pPos,
                       kv_offset_bytes=0):
    check(lib().q4_multi_head_attention(output.ptr, q.ptr, key_cache.ptr + kv_offset_bytes, value_cache.ptr + kv_offset_bytes,
                                        att.ptr if att else None, num_heads, head_size, kv_mul, max_seq_len, pPos.ptr))


def synchronize():
    check(lib().q4_stream_synchronize())


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int()
    mem = C.c_size_t()
    check(lib().q4_device_info(name, 256, C.byref(cu), C.byref(mem)))
    return name.value.decode(), cu.value, mem.value


class Transformer:
    """build_transformer / free_transformer (llama2_q4.cu:408-432) + run_transformer + sampler, by handle."""

    def __init__(self, path, perplexity=False, temperature=0.0, topp=0.9, seed=1, quiet=True):
        L = lib()
        L.q4_set_quiet(1 if quiet else 0)
        st = C.c_int()
        self.h = L.q4_transformer_new(path.encode(), int(perplexity), C.byref(st))
        if not self.h:
            raise Q4Error("build_transformer failed: %s %s" % (L.q4_status_string(st.value).decode(), L.q4_last_error().decode()))
        # a COPY of the header: the C struct lives inside the Transformer and dies with close(); readers of
        # `config` (bench.py prints it after closing) must never see freed memory
        self.config = Config.from_buffer_copy(L.q4_transformer_config(self.h).contents)
        self.state = L.q4_transformer_state(self.h)
        self.weights = L.q4_transformer_weights(self.h)
        self.sampler = L.q4_sampler_new(self.config.vocab_size, temperature, topp, seed)
        if not self.sampler:
            raise Q4Error("build_sampler failed")

    def close(self):
        L = lib()
        if getattr(self, "h", None):
            L.q4_stream_synchronize()
            L.q4_reset_graphs()
            L.q4_sampler_delete(self.sampler)
            L.q4_transformer_delete(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, prompt_tokens):
        t = np.ascontiguousarray(prompt_tokens, dtype=np.int32)
        check(lib().q4_reset_sequence(self.state, t.ctypes.data, t.shape[0]))

    def run_transformer(self, gen_token, copy_logits=False):
        check(lib().q4_run_transformer(int(gen_token), C.byref(self.config), self.state, self.weights, int(copy_logits),
                                       self.sampler))

    def logits(self):
        out = np.empty(self.config.vocab_size, dtype=np.float16)
        check(lib().q4_get_logits(self.h, out.ctypes.data))
        return out

    def kv_row(self, layer, pos):
        kv_dim = self.config.dim * self.config.n_kv_heads // self.config.n_heads
        k = np.empty(kv_dim, dtype=np.float16)
        v = np.empty(kv_dim, dtype=np.float16)
        check(lib().q4_get_kv_row(self.h, layer, pos, k.ctypes.data, v.ctypes.data))
        return k, v

    def logits_array(self, num_pos):
        out = np.empty((num_pos, self.config.vocab_size), dtype=np.float32)
        check(lib().q4_get_logits_array(self.h, num_pos, out.ctypes.data))
        return out

    def pos(self):
        return lib().q4_shared_pos(self.state)

    def token(self, i):
        return lib().q4_shared_token(self.state, i)

    def generate_ids(self, prompt_tokens, steps):
        """generate() on token ids: returns (tokens ring [pos+1], tok/s, timed_tokens, seconds)."""
        t = np.ascontiguousarray(prompt_tokens, dtype=np.int32)
        if steps <= 0 or steps > self.config.seq_len:          # the C side clamps the same way (llama2_q4.cu:690)
            steps = self.config.seq_len
        out = np.zeros(steps + 2, dtype=np.int32)
        timed = C.c_int()
        secs = C.c_double()
        tps = lib().q4_generate_ids(self.h, self.sampler, t.ctypes.data, t.shape[0], steps, out.ctypes.data, C.byref(timed),
                                    C.byref(secs))
        if tps < 0:
            raise Q4Error("generate failed: " + lib().q4_last_error().decode())
        return out[: timed.value + 2], tps, timed.value, secs.value

    def perplexity_ids(self, tokens_with_bos):
        t = np.ascontiguousarray(tokens_with_bos, dtype=np.int32)
        return lib().q4_perplexity_ids(self.h, self.sampler, t.ctypes.data, t.shape[0] - 1)

    def bench_kernel(self, kernel_id, iters):
        mn = C.c_double()
        mx = C.c_double()
        avg = lib().q4_bench_kernel(kernel_id, C.byref(self.config), self.state, self.weights, iters, C.byref(mn), C.byref(mx))
        if avg < 0:
            raise Q4Error("bench_kernel failed: " + lib().q4_last_error().decode())
        return avg, mn.value, mx.value


    def bench_in_network(self, report_mask, tokens=8, time_mask=127):
        """(avg, min, max us, launches) of one launch class inside the eager decode network, from the current position."""
        mn = C.c_double()
        mx = C.c_double()
        n = C.c_int()
        avg = lib().q4_bench_in_network(time_mask, report_mask, C.byref(self.config), self.state, self.weights, tokens, C.byref(mn),
                                        C.byref(mx), C.byref(n))
        if avg < 0:
            raise Q4Error("bench_in_network failed: " + lib().q4_last_error().decode())
        return avg, mn.value, mx.value, n.value

    def bench_kernel_graph(self, kernel_id, iters=32, reps=20):
        us = lib().q4_bench_kernel_graph(kernel_id, C.byref(self.config), self.state, self.weights, iters, reps)
        if us < 0:
            raise Q4Error("bench_kernel_graph failed: " + lib().q4_last_error().decode())
        return us


class Tokenizer:
    def __init__(self, path, vocab_size):
        self.h = lib().q4_tokenizer_new(path.encode(), vocab_size)
        if not self.h:
            raise Q4Error("couldn't load %s" % path)

    def encode(self, text, bos=1, eos=0):
        b = text.encode("utf-8") if isinstance(text, str) else text
        toks = np.zeros(len(b) + 3, dtype=np.int32)
        n = C.c_int()
        check(lib().q4_tokenizer_encode(self.h, b, bos, eos, toks.ctypes.data, C.byref(n)))
        return toks[: n.value].tolist()

    def decode(self, prev, tok):
        return lib().q4_tokenizer_decode(self.h, prev, tok)

    def close(self):
        if self.h:
            lib().q4_tokenizer_delete(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
