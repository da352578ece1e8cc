"""Offline format pipeline: own weight_packer vs the reference's (bytes, padding nibbles masked) and
the converter counterpart vs the reference convert_awq_to_bin.py (sha256 manifest)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

import packer_util
from conftest import GOLDEN, ROOT

PACKER = os.path.join(ROOT, "llama_cu_awq_amd", "bin", "weight_packer")
REF_PACKER = os.path.join(ROOT, "oracle", "_ref", "weight_packer")


@pytest.mark.parametrize("fmt", [1, 0])
def test_own_packer_matches_reference_golden(tmp_path, fmt):
    g = json.load(open(os.path.join(GOLDEN, "packer_goldens.json")))["old_format_%d" % fmt]
    cfg = packer_util.write_awq_dump(str(tmp_path), old_format=bool(fmt), seed=99)
    dst = str(tmp_path / "out.bin")
    subprocess.check_call([PACKER, str(tmp_path / "config.json"), str(tmp_path / "dump"), dst, str(fmt)], stdout=subprocess.DEVNULL)
    data = open(dst, "rb").read()
    assert len(data) == g["bytes"] == 362528
    assert hashlib.sha256(packer_util.mask_zero_padding(data, cfg)).hexdigest() == g["sha256_masked"]


@pytest.mark.skipif(not os.path.exists(REF_PACKER), reason="oracle/_ref/weight_packer not built (needs /root/reference)")
@pytest.mark.parametrize("fmt,seed", [(1, 5), (0, 6)])
def test_own_packer_matches_reference_binary(tmp_path, fmt, seed):
    cfg = dict(packer_util.CFG, hidden_size=384, intermediate_size=256, num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2)
    cfg = packer_util.write_awq_dump(str(tmp_path), old_format=bool(fmt), seed=seed, cfg=cfg)   # GQA + no rope_theta fallback path
    outs = []
    for exe in (REF_PACKER, PACKER):
        dst = str(tmp_path / (os.path.basename(os.path.dirname(exe)) + ".bin"))
        subprocess.check_call([exe, str(tmp_path / "config.json"), str(tmp_path / "dump"), dst, str(fmt)], stdout=subprocess.DEVNULL)
        outs.append(packer_util.mask_zero_padding(open(dst, "rb").read(), cfg))
    assert outs[0] == outs[1]


def test_packer_output_loads(tmp_path, orc):
    """The packed file is a valid checkpoint for the loader (header + exact size)."""
    packer_util.write_awq_dump(str(tmp_path), old_format=True, seed=1)
    dst = str(tmp_path / "out.bin")
    subprocess.check_call([PACKER, str(tmp_path / "config.json"), str(tmp_path / "dump"), dst, "1"], stdout=subprocess.DEVNULL)
    m = orc.Model(dst)
    assert (m.cfg.dim, m.cfg.hidden_dim, m.cfg.n_layers, m.cfg.n_heads, m.cfg.n_kv_heads, m.cfg.vocab_size, m.cfg.seq_len) == (256, 384, 1, 2, 2, 64, 32)
    m.close()


def test_converter_matches_reference_manifest(tmp_path):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_awq_to_bin
    g = json.load(open(os.path.join(GOLDEN, "convert_goldens.json")))
    pt = str(tmp_path / "sd.pt")
    torch.save(packer_util.synthetic_state_dict(seed=4), pt)
    convert_awq_to_bin.convert(pt, str(tmp_path / "out"))
    got = {fn: hashlib.sha256(open(str(tmp_path / "out" / fn), "rb").read()).hexdigest() for fn in sorted(os.listdir(str(tmp_path / "out")))}
    assert got == g and len(got) == 28
