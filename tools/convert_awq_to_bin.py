#!/usr/bin/env python3
"""Counterpart of the reference's convert_awq_to_bin.py (lines 1-31): dump every tensor of a torch state-dict
to `<out_dir>/<key>.bin` (raw little-endian bytes of `value.cpu().numpy()`), ready for weight_packer.

usage: convert_awq_to_bin.py <checkpoint.pt|.bin> <out_dir>
"""
import os
import sys


def convert(checkpoint, out_dir):
    import torch

    state = torch.load(checkpoint, map_location="cpu")
    os.makedirs(out_dir, exist_ok=True)
    written = []
    if isinstance(state, dict):
        for key, value in state.items():
            print(key, type(value))
            if isinstance(value, torch.Tensor):
                print(value.shape, value.dtype)
                path = os.path.join(out_dir, key + ".bin")
                with open(path, "wb") as f:
                    f.write(value.cpu().numpy().tobytes())
                written.append(path)
    return written


if __name__ == "__main__":
    if len(sys.argv) != 3:
        print(__doc__)
        sys.exit(1)
    convert(sys.argv[1], sys.argv[2])
