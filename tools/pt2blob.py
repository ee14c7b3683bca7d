#!/usr/bin/env python3
"""Convert a checkpoint written by the reference trainer (`torch::save(net, path)`,
util.h:108-139: a libtorch module archive) into the flat fp32 blob the HIP tower
loads (`<path>.azw`, read by az/util.hpp and alphazero_amd.weights).

    python tools/pt2blob.py model_12.pt [--board 15]

Offline tool: libtorch/PyTorch is used here only, never on the hot path.
"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import synth  # noqa: E402


def load_state(path):
    import torch
    m = torch.jit.load(path, map_location="cpu")      # module archive: parameters/buffers, no code is run
    sd = {k: v.detach().cpu().float().numpy() for k, v in m.state_dict().items()}
    return sd


def state_to_blob(sd):
    Cs = [sd["CBlock.conv.weight"].shape[0]]
    i = 0
    while "RBlock_%d.conv1.weight" % i in sd:
        Cs.append(sd["RBlock_%d.conv1.weight" % i].shape[0])
        i += 1
    c_in = sd["CBlock.conv.weight"].shape[1]
    c_out = sd["PHead.conv.weight"].shape[0]
    A = sd["VHead.fc1.weight"].shape[1]
    N = int(round(A ** 0.5))
    parts = []
    for name, shape in synth.blob_layout(N, c_in, Cs, c_out):
        a = sd[name]
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        parts.append(a.reshape(-1).astype(np.float32))
    return N, c_in, c_out, Cs, np.concatenate(parts)


def write_azw(path, N, c_in, c_out, Cs, blob):
    with open(path, "wb") as f:
        f.write(b"AZW1")
        f.write(struct.pack("<4I", N, c_in, c_out, len(Cs)))
        f.write(struct.pack("<%dI" % len(Cs), *Cs))
        f.write(struct.pack("<Q", blob.size))
        f.write(np.ascontiguousarray(blob, np.float32).tobytes())


def read_azw(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"AZW1"
        N, c_in, c_out, n = struct.unpack("<4I", f.read(16))
        Cs = list(struct.unpack("<%dI" % n, f.read(4 * n)))
        (cnt,) = struct.unpack("<Q", f.read(8))
        blob = np.frombuffer(f.read(4 * cnt), np.float32).copy()
    return N, c_in, c_out, Cs, blob


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("-o", "--out")
    a = ap.parse_args()
    N, c_in, c_out, Cs, blob = state_to_blob(load_state(a.checkpoint))
    out = a.out or a.checkpoint + ".azw"
    write_azw(out, N, c_in, c_out, Cs, blob)
    print("wrote %s: N=%d c_in=%d c_out=%d Cs=%s floats=%d" % (out, N, c_in, c_out, Cs, blob.size))
