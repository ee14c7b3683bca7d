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


class TrainerBuffer:
    """The trainer-side deque of replay.hpp, to predict what `train` holds after it loaded a
    `replay/` directory and received rows: rows are (state, policy, reward) tuples.

    The two ways rows enter evict differently:
      * `receive` (replay.hpp:152-159, one zmq message): append, then signal when the size EQUALS
        the threshold, then pop the oldest row while size > max_size   -> keeps max_size rows;
      * `load` (replay.hpp:190-206, rows of the saved tensors): signal BEFORE the append when the
        size equals the threshold, append, then pop while size >= max_size -> keeps max_size - 1.
    """

    def __init__(self, max_size=1, threshold=0):
        self.max_size, self.threshold = int(max_size), int(threshold)
        self.rows = []
        self.above_threshold = False

    def receive(self, row):
        self.rows.append(row)
        if len(self.rows) == self.threshold:
            self.above_threshold = True
        if len(self.rows) > self.max_size:
            self.rows.pop(0)

    def load(self, rows):
        for row in rows:
            if len(self.rows) == self.threshold:
                self.above_threshold = True
            self.rows.append(row)
            if len(self.rows) >= self.max_size:
                self.rows.pop(0)

    def load_dir(self, directory):
        got = load_replay(directory)
        if got is None:
            return False
        state, policy, reward = got
        self.load(list(zip(state, policy, reward)))
        return True

    def save_dir(self, directory):
        """replay.hpp:241-274: the three tensors, rows in deque order."""
        if not self.rows:
            return
        save_replay(directory, np.stack([r[0] for r in self.rows]), np.stack([r[1] for r in self.rows]),
                    np.stack([r[2] for r in self.rows]))
