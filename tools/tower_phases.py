#!/usr/bin/env python3
"""Phase timing of the fused tower kernel (needs tools/dbg/libazhip_timing.so, built from
tower.hip with -DAZ_TOWER_TIMING): shader-clock stamps of wave 0 in the first and the last
workgroup at setup / per-layer tap loop / per-layer epilogue / heads boundaries."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "tools", "dbg", os.environ.get("AZ_TIMING_LIB", "libazhip_timing.so"))   # tools/dbg is git-ignored: build it first (see docstring)


def main():
    N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 15, int(sys.argv[2]) if len(sys.argv) > 2 else 512
    Cs = synth.NETCONFIG2
    eng = capi.Engine(N, 5, B, Cs=Cs, max_budget=1, node_cap=8)
    eng.load_weights(synth.make_blob(N, 3, Cs, 0))
    boards = synth.random_boards(B, N, seed=0, occupancy=0.3)
    for _ in range(3):
        eng.tower_forward(boards)
    ts = np.zeros(128, np.uint64)
    assert capi.load().az_tower_timing_read(ts.ctypes.data_as(C.c_void_p)) == 0
    ts = ts.reshape(2, 64).astype(np.int64)
    L = 2 * (len(Cs) - 1)
    for blk in range(2):
        t = ts[blk]
        real_us = (t[62] - t[63]) / 100.0           # 100 MHz constant clock
        total = t[60] - t[0]
        taps = sum(t[2 + 2 * l] - (t[1] if l == 0 else t[3 + 2 * (l - 1)]) for l in range(L))
        epi = sum(t[3 + 2 * l] - t[2 + 2 * l] for l in range(L))
        print("block %s: total %d ticks = %.1f us (%.3f ticks/ns); setup+stem %d, tap loops %d (%.0f per layer), "
              "epilogues %d (%.0f per layer), heads %d" %
              ("first" if blk == 0 else "last", total, real_us, total / (real_us * 1e3), t[1] - t[0], taps, taps / L,
               epi, epi / L, t[60] - t[3 + 2 * (L - 1)]))
        print("   setup: zero+loads %d, im2col %d, stem mfma %d, stem epilogue+first weights %d; heads: 1x1 convs %d, softmax %d, value mlp %d" %
              (t[40] - t[0], t[41] - t[40], t[42] - t[41], t[1] - t[42], t[43] - t[3 + 2 * (L - 1)], t[44] - t[43], t[60] - t[44]))
        if blk == 0:
            print("   layer 8 tap starts (delta):", [int(t[46 + k] - t[45 + k]) for k in range(8)], "last tap:", int(t[2 + 2 * 8] - t[53]),
                  "prologue:", int(t[45] - t[3 + 2 * 7]))
        print("   per-layer taps:", [int(t[2 + 2 * l] - (t[1] if l == 0 else t[3 + 2 * (l - 1)])) for l in range(L)])
        print("   per-layer epi :", [int(t[3 + 2 * l] - t[2 + 2 * l]) for l in range(L)])


if __name__ == "__main__":
    main()
