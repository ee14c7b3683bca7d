#!/usr/bin/env python3
"""Numerical experiment (CPU, torch): what would a Winograd F(2x2,3x3) tower cost in accuracy?

Runs the NetConfig-2 tower on random boards four ways and prints the maximum absolute error of
policy and value against the fp32 forward:
  direct/16 : 3x3 convolutions with activations and weights rounded to 16 bits, fp32 accumulation,
              fp32 residual stream -- the arithmetic of the shipped MFMA kernels;
  wino/16   : the same with every 3x3 convolution as Winograd F(2x2,3x3): input tiles transformed
              in fp32 and rounded to 16 bits, weights transformed in fp64 and rounded to 16 bits,
              16 element-wise GEMMs accumulated in fp32, output transform in fp32.
Both for bf16 and fp16 operands, for random-init weights and for sharpened heads.

    python tools/winograd_error.py [--boards 64] [--board 15]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import synth  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def rnd(x, dt):
    return x if dt is None else x.to(dt).to(torch.float32)


def conv_direct(x, w, dt):
    return F.conv2d(rnd(x, dt), rnd(w, dt), padding=1)


def conv_wino(x, w, dt):
    B, C, N, _ = x.shape
    T = (N + 1) // 2                                   # 2x2 output tiles per side
    xp = F.pad(x, (1, 1 + 2 * T - N, 1, 1 + 2 * T - N))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)         # [B,C,T,T,4,4]
    bt = BT.float()
    V = rnd(torch.einsum("ij,bcxyjk,lk->bcxyil", bt, tiles, bt), dt)          # input transform
    U = rnd(torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G).float(), dt)     # weight transform
    M = torch.einsum("ocil,bcxyil->boxyil", U, V)                              # 16 GEMMs, fp32 accumulate
    at = AT.float()
    Y = torch.einsum("ij,boxyjk,lk->boxyil", at, M, at)                        # [B,O,T,T,2,2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], 2 * T, 2 * T)
    return Y[:, :, :N, :N]


def fold(w, b, bn):
    g, beta, rm, rv = [torch.from_numpy(np.array(t)).double() for t in bn]
    s = g / torch.sqrt(rv + 1e-5)
    return (torch.from_numpy(np.array(w)).double() * s[:, None, None, None]).float(), \
           ((torch.from_numpy(np.array(b)).double() - rm) * s + beta).float()


def forward(P, boards, Cs, conv, dt):
    N = boards.shape[-1]
    bn = lambda n: [P[n + "." + s] for s in ("weight", "bias", "running_mean", "running_var")]  # noqa: E731
    x = torch.from_numpy(boards).float()
    w, b = fold(P["CBlock.conv.weight"], P["CBlock.conv.bias"], bn("CBlock.bn"))
    x = F.relu(conv_direct(x, w, dt) + b[None, :, None, None])                 # stem stays direct (K = 27)
    for i in range(len(Cs) - 1):
        r = "RBlock_%d" % i
        w1, b1 = fold(P[r + ".conv1.weight"], P[r + ".conv1.bias"], bn(r + ".bn1"))
        w2, b2 = fold(P[r + ".conv2.weight"], P[r + ".conv2.bias"], bn(r + ".bn2"))
        h = F.relu(conv(x, w1, dt) + b1[None, :, None, None])
        x = F.relu(conv(h, w2, dt) + b2[None, :, None, None] + x)              # residual stream stays fp32
    pw = torch.from_numpy(np.array(P["PHead.conv.weight"])).float()
    pol = F.softmax((F.conv2d(x, pw) + float(P["PHead.conv.bias"][0])).flatten(1), dim=1)
    vw, vb = fold(P["VHead.conv.weight"], P["VHead.conv.bias"], bn("VHead.bn"))
    v = F.relu(F.conv2d(x, vw) + vb[None, :, None, None]).flatten(1)
    v = F.relu(v @ torch.from_numpy(np.array(P["VHead.fc1.weight"])).float().T + torch.from_numpy(np.array(P["VHead.fc1.bias"])).float())
    v = v @ torch.from_numpy(np.array(P["VHead.fc2.weight"])).float().T + torch.from_numpy(np.array(P["VHead.fc2.bias"])).float()
    return pol, F.softmax(v, dim=1) * 2 - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=64)
    ap.add_argument("--board", type=int, default=15)
    a = ap.parse_args()
    N, Cs = a.board, synth.NETCONFIG2
    boards = synth.random_boards(a.boards, N, seed=3)
    torch.set_grad_enabled(False)
    for label, gains in (("random init", (1.0, 1.0)), ("heads x20", (20.0, 20.0))):
        P = synth.unpack_blob(synth.make_blob(N, 3, Cs, 0, phead_gain=gains[0], fc2_gain=gains[1]), N, 3, Cs)
        p0, v0 = forward(P, boards, Cs, conv_direct, None)
        pw, vw = forward(P, boards, Cs, conv_wino, None)
        print("%-12s policy max %.3f | fp32 winograd vs fp32 direct: policy %.2e value %.2e"
              % (label, float(p0.max()), float((pw - p0).abs().max()), float((vw - v0).abs().max())))
        for dname, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            for cname, cv in (("direct", conv_direct), ("wino", conv_wino)):
                p, v = forward(P, boards, Cs, cv, dt)
                print("    %-4s %-6s policy %.2e  value %.2e" % (dname, cname, float((p - p0).abs().max()), float((v - v0).abs().max())))


if __name__ == "__main__":
    main()
