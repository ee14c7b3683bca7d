#!/usr/bin/env python3
"""Phase timing of the per-layer heads kernel (needs tools/dbg/libazhip_timing.so, built from tower.hip with
-DAZ_TOWER_TIMING): 100 MHz stamps of thread 0 of block 0 at the kernel's phase boundaries, on the Blokus tower's
shape (20x20, 128 channels, 91 policy channels, 4 values) or the shape given as N c_out [boards].  AZ_TIMING_LIB names
the library (an absolute path, or a file under tools/dbg); AZ_HEADS_MASKED=1 times the masked form with 300 wanted
entries per board."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "tools", "dbg", os.environ.get("AZ_TIMING_LIB", "libazhip_timing.so"))


def main():
    import torch
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    c_out = int(sys.argv[2]) if len(sys.argv) > 2 else 91
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    c_in, vd = (5, 4) if c_out > 1 else (3, 2)
    Cs = synth.NETCONFIG2
    eng = capi.Engine(N, 5, B, Cs=Cs, max_budget=1, node_cap=8, c_in=c_in, c_out=c_out, value_dim=vd)
    eng.load_weights(synth.make_blob(N, c_in, Cs, 0, c_out=c_out, value_dim=vd))
    planes = torch.from_numpy(synth.random_planes(B, c_in, N, seed=0) if c_in != 3 else synth.random_boards(B, N, seed=0)).cuda()
    pol = torch.empty(B, c_out * N * N, dtype=torch.float32, device="cuda")
    val = torch.empty(B, vd, dtype=torch.float32, device="cuda")
    masked = os.environ.get("AZ_HEADS_MASKED") == "1"          # ~300 wanted entries per board, as a Blokus expansion asks for
    rng = np.random.default_rng(0)
    words = np.zeros((B, c_out * N), np.uint32)
    for b in range(B if masked else 0):
        idx = rng.choice(c_out * N * N, 300, replace=False)
        np.bitwise_or.at(words[b], idx // N, (1 << (idx % N)).astype(np.uint32))
    wanted = torch.from_numpy(words.view(np.int32)).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        if masked:
            eng.tower_forward_masked_dev(planes.data_ptr(), B, pol.data_ptr(), val.data_ptr(), wanted.data_ptr())
        else:
            eng.tower_forward_dev(planes.data_ptr(), B, pol.data_ptr(), val.data_ptr())
    eng.sync()
    ts = np.zeros(32, np.uint64)
    lib = capi.load()
    lib.az_heads_timing_read.argtypes = [C.c_void_p]
    assert lib.az_heads_timing_read(ts.ctypes.data_as(C.c_void_p)) == 0
    t = ts.astype(np.int64)
    us = lambda a, b: (t[b] - t[a]) / 100.0          # noqa: E731
    print("masked" if masked else "dense", "heads block 0: total %.2f us; weight image + masks %.2f" % (us(0, 30), us(0, 1)))
    for i in range(6):
        if t[4 + 3 * i]:
            print("  unit %d%s of wave 0: bias/mask + mfma issue %.2f, epilogue %.2f%s" %
                  (i, "+" if i == 5 else "", us(2 + 3 * i, 3 + 3 * i), us(3 + 3 * i, 4 + 3 * i),
                   "" if i == 5 or not t[2 + 3 * (i + 1)] else ", to the next unit %.2f" % us(4 + 3 * i, 2 + 3 * (i + 1))))
    print("  units done at %.2f; wait for the other waves %.2f, reductions %.2f, fc1 %.2f, softmax pass %.2f, fc2 %.2f" %
          (us(0, 25), us(25, 26), us(26, 27), us(27, 28), us(28, 29), us(29, 30)))

if __name__ == "__main__":
    main()
