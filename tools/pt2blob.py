#!/usr/bin/env python3
"""Convert a checkpoint written by the reference trainer (`torch::save(net, path)`,
util.h:108-139: a libtorch module archive) into the flat fp32 blob the HIP tower
loads (`<path>.azw`, read by az/util.hpp and alphazero_amd.weights).

    python tools/pt2blob.py model_12.pt
    python tools/pt2blob.py --watch ckpt_location.txt      # beside an unmodified trainer

With --watch it follows the trainer's pointer file (read under flock(LOCK_SH) like util.h:63-87) and
writes the `.azw` twin of every checkpoint the file names, so that a generator whose load() found a
`.pt` without its twin picks the new network up at its next load().

Offline tool: libtorch/PyTorch is used here only, never on the hot path.
"""
import argparse
import fcntl
import os
import struct
import sys
import time

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


def convert(checkpoint, out=None):
    """<checkpoint> -> <checkpoint>.azw, written under a temporary name and renamed: a reader never sees half a file."""
    N, c_in, c_out, Cs, blob = state_to_blob(load_state(checkpoint))
    out = out or checkpoint + ".azw"
    tmp = out + ".tmp%d" % os.getpid()
    write_azw(tmp, N, c_in, c_out, Cs, blob)
    os.replace(tmp, out)
    return "wrote %s: N=%d c_in=%d c_out=%d Cs=%s floats=%d" % (out, N, c_in, c_out, Cs, blob.size)


def read_pointer(pointer_file):
    """The path the trainer's pointer file names (util.h:63-87: first line, shared lock), or None."""
    try:
        with open(pointer_file) as f:
            fcntl.flock(f, fcntl.LOCK_SH)
            try:
                return f.readline().strip() or None
            finally:
                fcntl.flock(f, fcntl.LOCK_UN)
    except OSError:
        return None


def watch_once(pointer_file, convert_fn=convert):
    """One look at the pointer file: converts the named checkpoint if it exists and has no twin yet.
    Relative paths are taken from the pointer file's directory (the trainer's working directory)."""
    path = read_pointer(pointer_file)
    if not path or path.endswith(".azw"):
        return None
    full = path if os.path.isabs(path) else os.path.join(os.path.dirname(os.path.abspath(pointer_file)), path)
    if not os.path.exists(full) or os.path.exists(full + ".azw"):
        return None
    return convert_fn(full)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint", nargs="?")
    ap.add_argument("-o", "--out")
    ap.add_argument("--watch", metavar="POINTER_FILE", help="follow this ckpt_location.txt and convert what it names")
    ap.add_argument("--interval", type=float, default=1.0, help="seconds between looks at the pointer file")
    a = ap.parse_args()
    if a.watch:
        while True:
            try:
                msg = watch_once(a.watch)
                if msg:
                    print(msg, flush=True)
            except Exception as e:  # noqa: BLE001 -- a checkpoint still being written: look again
                print("pt2blob: %r, retrying" % (e,), flush=True)
            time.sleep(a.interval)
    if not a.checkpoint:
        ap.error("a checkpoint or --watch is needed")
    print(convert(a.checkpoint, a.out))
