#!/usr/bin/env python3
"""Fingerprints of the gfx950 ISA of every kernel of a translation unit: instruction count, scratch / VGPR / LDS figures and a hash of
the instruction stream with labels and symbol names normalised. Used to show that a source reorganisation left the product's kernels
unchanged (round 5: experiment code moved to csrc/exp/).   tools/isa_fingerprint.py file.hip [extra hipcc flags...] > fingerprints.json"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = sys.argv[2:]
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
        "-I" + os.path.join(ROOT, "llama_cu_awq_amd", "csrc"), "-S", "--cuda-device-only", "-o", "-"]
name = os.path.basename(src)
if name.startswith("gemv_") or name.startswith("layer_attn"):
    base += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
asm = subprocess.run(base + flags + [src], capture_output=True, text=True, check=True).stdout
out = {}
cur, body = None, []
meta = {}
for line in asm.splitlines():
    m = re.match(r"^(_Z\w+):\s*; @", line)
    if m:
        cur, body = m.group(1), []
        continue
    if cur is None:
        continue
    if line.startswith(".Lfunc_end"):
        text = "\n".join(body)
        text = re.sub(r"_Z\w+", "SYM", text)
        labels = {}
        def lab(mm):
            return labels.setdefault(mm.group(0), "L%d" % len(labels))
        text = re.sub(r"\.LBB\d+_\d+", lab, text)
        out[cur] = {"instructions": len(body), "sha": hashlib.sha256(text.encode()).hexdigest()[:16]}
        cur = None
        continue
    ins = line.split(";")[0].strip()
    if ins and not ins.startswith("."):
        body.append(ins)
    elif ins.startswith(".LBB"):
        body.append(ins)
for m in re.finditer(r"\.amdhsa_kernel (_Z\w+)(.*?)\.end_amdhsa_kernel", asm, re.S):
    k, blk = m.group(1), m.group(2)
    if k in out:
        for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
            mm = re.search(r"\.amdhsa_%s (\d+)" % key, blk)
            if mm:
                out[k][key] = int(mm.group(1))
dem = subprocess.run(["c++filt"], input="\n".join(out.keys()), capture_output=True, text=True).stdout.splitlines()
res = {}
for k, d in zip(out.keys(), dem):
    d = re.sub(r"\(.*$", "", d)          # the template name without the argument list
    res[d] = out[k]
json.dump(res, sys.stdout, indent=1, sort_keys=True)
