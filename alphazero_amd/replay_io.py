"""The trainer's replay files: `<dir>/replay.state|.policy|.reward`, three
tensors written with `torch::save(tensor, path)` (replay.hpp:241-274) and read
back with `torch::load` (replay.hpp:163-208): state u8 [n,C,N,N], policy f32
[n,c_out,N,N], reward f32 [n,2].

`torch::save(Tensor)` writes a TorchScript module archive whose only member is a
parameter named "0" (OutputArchive::write("0", tensor)), so the files are read
with torch.jit.load and written by scripting a one-parameter holder module.
Offline tooling: torch is never on the hot path.
"""
import os

import numpy as np


def _torch():
    import torch
    return torch


def load_tensor(path):
    torch = _torch()
    m = torch.jit.load(path, map_location="cpu")
    return dict(m.named_parameters())["0"].detach().cpu().numpy()


def _holder_class():
    torch = _torch()

    class Holder(torch.nn.Module):
        def __init__(self, t):
            super().__init__()
            self.register_parameter("0", torch.nn.Parameter(t, requires_grad=False))

        def forward(self):
            return None
    return Holder


def save_tensor(array, path):
    torch = _torch()
    t = torch.from_numpy(np.ascontiguousarray(array))
    torch.jit.script(_holder_class()(t)).save(path)


def load_replay(directory):
    """-> (state u8 [n,C,N,N], policy f32 [n,c_out,N,N], reward f32 [n,2]) or None (replay.hpp:173-177)."""
    p = os.path.join(directory, "replay")
    if not os.path.exists(p + ".state"):
        return None
    return load_tensor(p + ".state"), load_tensor(p + ".policy"), load_tensor(p + ".reward")


def save_replay(directory, state, policy, reward):
    os.makedirs(directory, exist_ok=True)
    p = os.path.join(directory, "replay")
    save_tensor(np.asarray(state, np.uint8), p + ".state")
    save_tensor(np.asarray(policy, np.float32), p + ".policy")
    save_tensor(np.asarray(reward, np.float32), p + ".reward")
