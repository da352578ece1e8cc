#!/usr/bin/env python3
"""Regenerates tests/golden/*.json|*.bin|*.npz. Run in the BUILD container (needs /root/reference for the pieces of
the reference that compile there: tokenizer.h behind oracle/ref_tokenizer_shim.cpp, and weight_packer.cpp; both are
built by `make -C oracle` into oracle/_ref/). Nothing here is read at test time except the files it writes.

  tokenizer_goldens.json : encode()/decode() of the REFERENCE tokenizer on a list of strings (tokenizer.h:68-223)
  packer_goldens.json    : sha256 of the reference weight_packer's output (padding nibbles of `zeros` masked,
                           SURVEY P7) on seeded synthetic AWQ dumps, old and new format
  convert_goldens.json   : sha256 of every file the reference convert_awq_to_bin.py writes for a seeded state-dict
  rng_goldens.json       : output of the reference's own random_u32 / random_f32 (sampler.h:31-40, cut out where the file
                           lies, compiled with gcc in a scratch directory)
  perplexity_goldens.json: output of the reference's own softmax / compute_perplexity (perplexity.h:3-51) on seeded logits
  micro_model.bin/.npz   : a 2-layer checkpoint (synth.py, seed 5) and the CPU restatement's logits / KV / greedy
                           tokens on it -- self-generated (the reference has no runnable GPU path here): they pin the
                           oracle against regressions and give the GPU tests a committed fixture.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from llama_cu_awq_amd import synth  # noqa: E402
import packer_util  # noqa: E402

STRINGS = ["write an essay about GPUs", "", "Hello", "Once upon a time", "[INST] Hi [/INST]", " leading space", "tabs\tand\nnewlines",
           "UTF-8: héllo wörld 你好 \U0001F600", "1234567890 + 42 = ?", "a", "  ", "The quick brown fox jumps over the lazy dog.",
           "[INST] <<SYS>>\nYou are helpful.\n<</SYS>>\n\nWhat is a GPU? [/INST]", "ÿþ raw-ish bytes \x01\x02", "<s> </s> <unk> <0x41>"]


def tokenizer_goldens():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtokenizer_ref.so"))
    L.ref_tok_build.restype = C.c_void_p
    L.ref_tok_build.argtypes = [C.c_char_p, C.c_int]
    L.ref_tok_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    L.ref_tok_decode.restype = C.c_char_p
    L.ref_tok_decode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    t = L.ref_tok_build(os.path.join(REF, "tokenizer.bin").encode(), 32000)
    out = {"vocab_size": 32000, "max_token_length": int(L.ref_tok_max_token_length(C.c_void_p(t))), "encode": [], "decode": []}
    for s in STRINGS:
        b = s.encode("utf-8")
        for bos, eos in ((1, 0), (0, 0), (1, 1)):
            buf = (C.c_int * (len(b) + 8))()
            n = L.ref_tok_encode(t, b, bos, eos, buf)
            out["encode"].append({"text": s, "bos": bos, "eos": eos, "tokens": list(buf[:n])})
    for prev, tok in [(1, 2436), (2436, 385), (1, 29871), (5, 15043), (1, 65), (0, 3), (7, 13), (1, 259), (9, 31999), (1, 1), (2, 2)]:
        out["decode"].append({"prev": prev, "token": tok, "piece_hex": L.ref_tok_decode(t, prev, tok).hex()})
    json.dump(out, open(os.path.join(HERE, "tokenizer_goldens.json"), "w"), indent=1)


def packer_goldens():
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "weight_packer")
    out = {}
    for fmt in (1, 0):
        with tempfile.TemporaryDirectory() as d:
            cfg = packer_util.write_awq_dump(d, old_format=bool(fmt), seed=99)
            dst = os.path.join(d, "out.bin")
            subprocess.check_call([ref_bin, os.path.join(d, "config.json"), os.path.join(d, "dump"), dst, str(fmt)], stdout=subprocess.DEVNULL)
            data = packer_util.mask_zero_padding(open(dst, "rb").read(), cfg)
            out["old_format_%d" % fmt] = {"bytes": len(data), "sha256_masked": hashlib.sha256(data).hexdigest(), "config": cfg}
    json.dump(out, open(os.path.join(HERE, "packer_goldens.json"), "w"), indent=1)


def convert_goldens():
    import torch
    out = {}
    with tempfile.TemporaryDirectory() as d:
        sd = packer_util.synthetic_state_dict(seed=4)
        pt = os.path.join(d, "sd.pt")
        torch.save(sd, pt)
        dst = os.path.join(d, "out")
        subprocess.check_call([sys.executable, os.path.join(REF, "convert_awq_to_bin.py"), pt, dst], stdout=subprocess.DEVNULL)
        for fn in sorted(os.listdir(dst)):
            out[fn] = hashlib.sha256(open(os.path.join(dst, fn), "rb").read()).hexdigest()
    json.dump(out, open(os.path.join(HERE, "convert_goldens.json"), "w"), indent=1)


def rng_goldens():
    """sampler.h as a whole needs the CUDA runtime headers, but random_u32 / random_f32 (sampler.h:31-40) are plain C: the two
    function definitions are cut out of the reference file WHERE IT LIES into a scratch translation unit (never kept), compiled
    with gcc and run -- the committed JSON is their output for seed 1."""
    import re
    src = open(os.path.join(REF, "sampler.h")).read()
    m = re.search(r"unsigned int random_u32\(unsigned long long\* state\) \{.*?\n\}\nfloat random_f32\(unsigned long long\* state\) \{.*?\n\}", src, re.S)
    assert m, "random_u32 / random_f32 not found in the reference's sampler.h"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "rng.c")
        open(c, "w").write("#include <stdio.h>\n" + m.group(0) + """
int main(void) {
    unsigned long long s = 1;
    printf("{\\"seed\\": 1, \\"u32\\": [");
    for (int i = 0; i < 3; i++) printf("%s%u", i ? ", " : "", random_u32(&s));
    s = 1;
    printf("], \\"f32\\": [");
    for (int i = 0; i < 3; i++) printf("%s%.8f", i ? ", " : "", random_f32(&s));
    printf("]}\\n");
    return 0;
}
""")
        exe = os.path.join(d, "rng")
        subprocess.check_call(["gcc", "-O1", "-o", exe, c])
        out = json.loads(subprocess.check_output([exe]).decode())
    json.dump(out, open(os.path.join(HERE, "rng_goldens.json"), "w"), indent=1)


def perplexity_goldens():
    """perplexity.h:3-51 (softmax + compute_perplexity, host math in plain C++) cut out of the reference file where it lies into a
    scratch translation unit, compiled with g++ and run on seeded logits; the committed JSON holds inputs' seed and its output."""
    import re
    src = open(os.path.join(REF, "perplexity.h")).read()
    m = re.search(r"void softmax\(float\* x, int size\) \{.*?\n\}\n.*?float compute_perplexity\(int\* tokens, float\* logits, int num_tokens, int vocab_size\) \{.*?\n\}", src, re.S)
    assert m, "softmax / compute_perplexity not found in the reference's perplexity.h"
    n, v, seed = 24, 1000, 20240229
    rng = np.random.default_rng(seed)
    logits = (rng.standard_normal((n, v)) * 3.0).astype(np.float32)
    tokens = rng.integers(0, v, size=n).astype(np.int32)
    with tempfile.TemporaryDirectory() as d:
        logits.tofile(os.path.join(d, "logits.bin"))
        tokens.tofile(os.path.join(d, "tokens.bin"))
        c = os.path.join(d, "ppl.cpp")
        open(c, "w").write("#include <stdio.h>\n#include <stdlib.h>\n#include <math.h>\n" + m.group(0) + """
int main(int argc, char** argv) {
    const int n = atoi(argv[1]), v = atoi(argv[2]);
    float* logits = (float*)malloc(sizeof(float) * n * v);
    int* tokens = (int*)malloc(sizeof(int) * n);
    FILE* f = fopen(argv[3], "rb"); if (fread(logits, sizeof(float), (size_t)n * v, f) != (size_t)n * v) return 1; fclose(f);
    f = fopen(argv[4], "rb"); if (fread(tokens, sizeof(int), n, f) != (size_t)n) return 1; fclose(f);
    const float p = compute_perplexity(tokens, logits, n, v);
    printf("{\\"perplexity\\": %.9g, \\"softmax_row0\\": [", p);
    for (int i = 0; i < 8; i++) printf("%s%.9g", i ? ", " : "", logits[i]);
    printf("]}\\n");
    return 0;
}
""")
        exe = os.path.join(d, "ppl")
        subprocess.check_call(["g++", "-O1", "-o", exe, c])
        out = json.loads(subprocess.check_output([exe, str(n), str(v), os.path.join(d, "logits.bin"), os.path.join(d, "tokens.bin")]).decode())
    out.update({"num_tokens": n, "vocab_size": v, "seed": seed, "logit_scale": 3.0})
    json.dump(out, open(os.path.join(HERE, "perplexity_goldens.json"), "w"), indent=1)


def micro_model():
    import oracle
    path = os.path.join(HERE, "micro_model.bin")
    synth.write_model(path, synth.GEOMETRIES["micro"], seed=5)
    m = oracle.Model(path)
    prompt = np.array([1, 9, 77, 30], dtype=np.int32)
    toks, logits = m.generate_greedy(prompt, 20, want_logits=True)
    k, v = m.kv()
    np.savez_compressed(os.path.join(HERE, "micro_model_expected.npz"), prompt=prompt, tokens=toks, logits=logits.astype(np.float16),
                        k=k[:, :20], v=v[:, :20])
    m.close()


if __name__ == "__main__":
    tokenizer_goldens()
    packer_goldens()
    convert_goldens()
    rng_goldens()
    perplexity_goldens()
    micro_model()
    print("goldens written to", HERE)
