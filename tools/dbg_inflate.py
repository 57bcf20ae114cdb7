#!/usr/bin/env python3
"""Scratch: the test corpus of tests/test_gpu_inflate.py through the device decoder, status codes and first mismatch per member."""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
from iteres_amd import engine as eng, synth
import test_gpu_inflate as T
rng = np.random.default_rng(11)
inf = eng.Inflater()
names = ["L0", "L1", "L6", "L9", "FIXED", "HUFF", "RLE"]
for di, data in enumerate(T.corpus(rng)):
    data = data[:60000]
    for si, (level, strategy) in enumerate(((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE))):
        comp = T.member(data, level, strategy) + synth.BGZF_EOF
        out, status = inf.inflate(comp)
        got = out.tobytes()
        if status.any() or got != data:
            mm = next((i for i in range(min(len(got), len(data))) if got[i] != data[i]), -1)
            print(di, names[si], "len", len(data), "status", status.tolist(), "first mismatch", mm, flush=True)
print("done")
