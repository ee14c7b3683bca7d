"""Deterministic synthetic tower weights and boards.

The weight blob is a flat fp32 vector in the reference network's module
registration order (network.hpp:106-114): CBlock, PHead, VHead, RBlock_0..;
inside a module conv.weight, conv.bias, then per BatchNorm weight, bias,
running_mean, running_var; VHead adds fc1.{weight,bias}, fc2.{weight,bias};
RBlock_i is conv1.{w,b}, conv2.{w,b}, bn1.{w,b,rm,rv}, bn2.{w,b,rm,rv}.

Values come from a counter-based splitmix64 stream so the same blob can be
regenerated anywhere from (shape, seed) without shipping megabytes of floats.
"""
import numpy as np

NETCONFIG2 = [128] * 10          # netconfig.hpp:54-55 -> 9 ResBlocks (network.hpp:105)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed, n, offset=0):
    """n doubles in [0,1) from counter offset..offset+n of stream `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
        r = _splitmix64(idx)
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def blob_layout(N, c_in, Cs, c_out=1, value_dim=2):
    """[(name, shape)] in blob order."""
    num_res = len(Cs) - 1
    Ch = Cs[num_res - 1]
    L = []

    def conv(n, ci, co, k):
        L.append((n + ".weight", (co, ci, k, k)))
        L.append((n + ".bias", (co,)))

    def bn(n, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            L.append((n + "." + s, (c,)))

    conv("CBlock.conv", c_in, Cs[0], 3)
    bn("CBlock.bn", Cs[0])
    conv("PHead.conv", Ch, c_out, 1)
    conv("VHead.conv", Ch, 1, 1)
    bn("VHead.bn", 1)
    L.append(("VHead.fc1.weight", (64, N * N)))
    L.append(("VHead.fc1.bias", (64,)))
    L.append(("VHead.fc2.weight", (value_dim, 64)))
    L.append(("VHead.fc2.bias", (value_dim,)))
    for i in range(num_res):
        r = "RBlock_%d" % i
        conv(r + ".conv1", Cs[i], Cs[i + 1], 3)
        conv(r + ".conv2", Cs[i + 1], Cs[i + 1], 3)
        bn(r + ".bn1", Cs[i + 1])
        bn(r + ".bn2", Cs[i + 1])
    return L


def blob_size(N, c_in, Cs, c_out=1, value_dim=2):
    return int(sum(int(np.prod(s)) for _, s in blob_layout(N, c_in, Cs, c_out, value_dim)))


def conv_layer_names(Cs):
    """The BatchNorm-followed 3x3 convolutions in forward order with their BatchNorm: stem, then conv1 / conv2 of
    every ResBlock."""
    names = [("CBlock.conv", "CBlock.bn")]
    for i in range(len(Cs) - 1):
        names += [("RBlock_%d.conv1" % i, "RBlock_%d.bn1" % i), ("RBlock_%d.conv2" % i, "RBlock_%d.bn2" % i)]
    return names


def identity_calibration(Cs):
    """bn_spread argument of make_blob that leaves the convolutions as drawn (what a calibration starts from)."""
    n = 1 + 2 * (len(Cs) - 1)
    return {"scale": np.ones((n, Cs[0]), np.float32), "shift": None, "vconv": (1.0, None)}


def make_blob(N, c_in, Cs, seed=0, phead_gain=1.0, fc2_gain=1.0, c_out=1, value_dim=2, bn_spread=None):
    """Random-init weights: conv/linear U(-1/sqrt(fan_in), +), BN statistics
    randomised so that folding is exercised; head gains sharpen the outputs.

    bn_spread = None: benign BatchNorm (running_var in [0.5, 1.5], weight in [0.8, 1.2]).
    bn_spread = dict: BatchNorm as trained networks have it -- running_var log-uniform over 1e-4 .. 1e2,
    running_mean in +-3, weight in 0.01 .. 3 with about 3 % exact zeros, bias in +-0.5 -- and every convolution
    in front of a BatchNorm rescaled per output channel (`scale` [layer][C], applied to the drawn weights) and
    given the bias `shift` [layer][C] (None: the drawn one), `vconv` = (scale, shift) for VHead.conv.  With the
    scales of a calibration (oracle/gen_golden.py: the pre-BatchNorm activations then have about the statistics
    the BatchNorm holds, as after training) the network is well-conditioned; with identity_calibration() the
    statistics are unrelated to the activations and the folded weights span six orders of magnitude."""
    parts = []
    off = 0
    cal = {}
    if bn_spread is not None:
        for li, (cn, _) in enumerate(conv_layer_names(Cs)):
            cal[cn] = (bn_spread["scale"][li], None if bn_spread["shift"] is None else bn_spread["shift"][li])
        cal["VHead.conv"] = bn_spread["vconv"]
    for name, shape in blob_layout(N, c_in, Cs, c_out, value_dim):
        n = int(np.prod(shape))
        u = uniform01(seed, n, off)
        off += n
        spread = bn_spread is not None and ".bn" in name
        if spread and name.endswith("running_var"):
            x = 10.0 ** (-4.0 + 6.0 * u)
        elif spread and name.endswith("running_mean"):
            x = (u - 0.5) * 6.0
        elif spread and name.endswith("weight"):
            x = 0.01 + 2.99 * u
            if not name.startswith("VHead"):
                x[uniform01(seed + 7777, n, off) < 0.03] = 0.0          # channels the training switched off
        elif spread and name.endswith("bias"):
            x = u - 0.5
        elif name.endswith("running_var"):
            x = 0.5 + u
        elif name.endswith("running_mean"):
            x = (u - 0.5) * 0.4
        elif ".bn" in name and name.endswith("weight"):
            x = 0.8 + 0.4 * u
        elif ".bn" in name and name.endswith("bias"):
            x = (u - 0.5) * 0.4
        else:
            if name.endswith("weight"):
                fan_in = int(np.prod(shape[1:]))
            else:  # bias of the layer whose weight precedes it
                fan_in = prev_fan_in
            bound = 1.0 / np.sqrt(fan_in)
            x = (2.0 * u - 1.0) * bound
            prev_fan_in = fan_in
            layer = name.rsplit(".", 1)[0]
            if layer in cal:
                sc, sh = cal[layer]
                if name.endswith("weight"):
                    x = (x.reshape(shape[0], -1) * np.asarray(sc, np.float64).reshape(-1, 1)).reshape(-1)
                elif sh is not None:
                    x = np.broadcast_to(np.asarray(sh, np.float64), (n,)).copy()
        if name == "PHead.conv.weight":
            x = x * phead_gain
        if name == "VHead.fc2.weight":
            x = x * fc2_gain
        if name == "VHead.bn.bias":
            x = np.full(n, 0.5)
        parts.append(x.astype(np.float32))
    return np.concatenate(parts)


def fixture_blob(z):
    """The weights a tower fixture (tests/golden/tower_*.npz) was computed with, regenerated from its seed and,
    for the fixtures with trained-network-like BatchNorm, the per-channel calibration the fixture carries."""
    N, Cs = int(z["N"]), [int(c) for c in z["Cs"]]
    c_in, c_out = (int(z["c_in"]), int(z["c_out"])) if "c_in" in z.files else (3, 1)
    cal = None
    if "calib_scale" in z.files:
        cal = {"scale": z["calib_scale"], "shift": z["calib_shift"], "vconv": (float(z["calib_vconv"][0]), float(z["calib_vconv"][1]))}
    return make_blob(N, c_in, Cs, int(z["wseed"]), float(z["phead_gain"]), float(z["fc2_gain"]), c_out=c_out, bn_spread=cal)


def unpack_blob(blob, N, c_in, Cs, c_out=1, value_dim=2):
    out = {}
    off = 0
    for name, shape in blob_layout(N, c_in, Cs, c_out, value_dim):
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    assert off == blob.size
    return out


def random_boards(B, N, seed=0, occupancy=0.3):
    """u8 planes [B,3,N,N]: stones of player 0 / player 1 / side to move."""
    A = N * N
    u = uniform01(seed + 1000, B * A).reshape(B, A)
    side = uniform01(seed + 2000, B) < 0.5
    p0 = u < occupancy / 2
    p1 = (u >= occupancy / 2) & (u < occupancy)
    planes = np.zeros((B, 3, A), np.uint8)
    planes[:, 0] = p0
    planes[:, 1] = p1
    planes[:, 2] = side[:, None]
    return planes.reshape(B, 3, N, N)


def random_planes(B, c_in, N, seed=0, occupancy=0.35):
    """u8 planes [B,c_in,N,N] for a tower with c_in input planes: c_in - 1 disjoint stone planes and a last
    plane filled with the player to move (0 .. c_in - 2), the shape of the Blokus planes."""
    A = N * N
    u = uniform01(seed + 3000, B * A).reshape(B, A)
    side = (uniform01(seed + 4000, B) * (c_in - 1)).astype(np.uint8)
    planes = np.zeros((B, c_in, A), np.uint8)
    owners = c_in - 1
    for p in range(owners):
        planes[:, p] = (u >= occupancy * p / owners) & (u < occupancy * (p + 1) / owners)
    planes[:, c_in - 1] = side[:, None]
    return planes.reshape(B, c_in, N, N)
