#!/usr/bin/env python3
"""Where the fp16-operand tower's error against the fp32 reference comes from: a torch (CPU, float64) emulation of
the kernels' arithmetic -- BatchNorm folded in double, weights and the inputs of every 3x3 convolution rounded to
fp16, fp32 residual stream, exact heads -- with each rounding switched off in turn, on a tower fixture.

    python tools/fp16_error_ablation.py tests/golden/tower_n15_c256x21_sharp.npz [boards]
"""
import sys
import os
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import synth  # noqa: E402


def r16(t):
    return t.to(torch.float16).to(torch.float64)


def forward(w, Cs, boards, round_w=True, round_x=True, round_h=True, only_blocks=None, round_stem=False):
    dt = torch.float64
    x = torch.from_numpy(boards.astype(np.float64))

    def fold(cn, bnn):
        s = w[bnn + ".weight"] / torch.sqrt(w[bnn + ".running_var"] + 1e-5)
        W = w[cn + ".weight"] * s.view(-1, 1, 1, 1)
        b = (w[cn + ".bias"] - w[bnn + ".running_mean"]) * s + w[bnn + ".bias"]
        return W, b.to(torch.float32).to(dt)
    W, b = fold("CBlock.conv", "CBlock.bn")
    x = F.relu(F.conv2d(x, r16(W) if round_stem else W.to(torch.float32).to(dt), b, padding=1)).to(torch.float32).to(dt)
    for i in range(len(Cs) - 1):
        on = only_blocks is None or i in only_blocks
        r = "RBlock_%d" % i
        W1, b1 = fold(r + ".conv1", r + ".bn1")
        W2, b2 = fold(r + ".conv2", r + ".bn2")
        if round_w and on:
            W1, W2 = r16(W1), r16(W2)
        xi = r16(x) if (round_x and on) else x
        h = F.relu(F.conv2d(xi, W1, b1, padding=1))
        h = r16(h) if (round_h and on) else h
        x = F.relu(F.conv2d(h, W2, b2, padding=1) + x).to(torch.float32).to(dt)
    p = F.conv2d(x, w["PHead.conv.weight"], w["PHead.conv.bias"]).flatten(1).softmax(1)
    vs = w["VHead.bn.weight"] / torch.sqrt(w["VHead.bn.running_var"] + 1e-5)
    v = F.conv2d(x, w["VHead.conv.weight"] * vs.view(-1, 1, 1, 1), (w["VHead.conv.bias"] - w["VHead.bn.running_mean"]) * vs + w["VHead.bn.bias"])
    v = F.relu(F.linear(F.relu(v).flatten(1), w["VHead.fc1.weight"], w["VHead.fc1.bias"]))
    v = F.linear(v, w["VHead.fc2.weight"], w["VHead.fc2.bias"]).softmax(1) * 2 - 1
    return np.concatenate([p.numpy(), v.numpy()], axis=1)


def main():
    z = np.load(sys.argv[1])
    B = int(sys.argv[2]) if len(sys.argv) > 2 else min(int(z["B"]), 64)
    N, Cs = int(z["N"]), [int(c) for c in z["Cs"]]
    blob = synth.fixture_blob(z)
    w = {k: torch.from_numpy(v.astype(np.float64)) for k, v in synth.unpack_blob(blob, N, 3, Cs).items()}
    boards = z["boards"][:B]
    ref = np.concatenate([z["policy"][:B], z["value"][:B]], axis=1).astype(np.float64)
    A = N * N
    torch.set_num_threads(8)
    with torch.no_grad():
        def err(**kw):
            d = np.abs(forward(w, Cs, boards, **kw) - ref)
            e, ev = d[:, :A].max(1), d[:, A:].max(1)
            return "policy max %.3g  99%% %.3g  median %.3g  rms %.3g | value max %.3g rms %.3g" % (
                e.max(), np.quantile(e, 0.99), np.median(e), np.sqrt((e ** 2).mean()), ev.max(), np.sqrt((ev ** 2).mean()))
        print("exact (fold only)          ", err(round_w=False, round_x=False, round_h=False))
        print("all roundings (per-layer)  ", err())
        print("... and fp16 stem (fused)  ", err(round_stem=True))
        print("only the stem's weights    ", err(round_w=False, round_x=False, round_h=False, round_stem=True))
        print("weights exact              ", err(round_w=False))
        print("stream copies exact        ", err(round_x=False))
        print("hidden activations exact   ", err(round_h=False))
        print("only weights rounded       ", err(round_x=False, round_h=False))
        nb = len(Cs) - 1
        print("first half of blocks only  ", err(only_blocks=set(range(nb // 2))))
        print("second half of blocks only ", err(only_blocks=set(range(nb // 2, nb))))


if __name__ == "__main__":
    main()
