"""CPU-side checks of the C ABI: the library loads and exports every symbol
include/az_hip.h declares.  No compute call is made (no GPU here)."""
import os
import re

import numpy as np

from alphazero_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "az_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = capi.load()
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(L, name), name
    assert sorted(capi.EXPORTS) == decl


def test_replay_symmetries_host_helper_matches_oracle_and_torch():
    import torch
    import azo
    N = 5
    rng = np.random.default_rng(0)
    cells = rng.integers(0, 3, N * N).astype(np.uint8)
    pol = rng.random((N, N)).astype(np.float32)
    so, po = capi.replay_symmetries(N, cells, 1, pol)
    planes = azo.env_planes(N, cells, 1)
    so2, po2 = azo.replay_symmetries(N, planes, pol)
    assert np.array_equal(so, so2) and np.array_equal(po, po2)
    # replay.hpp:80-112 with torch itself: rot90(r,{1,2}) then .flip(1)
    st, pt = torch.from_numpy(planes), torch.from_numpy(pol)[None]
    v = 0
    for r in range(4):
        sr, pr = st.rot90(r, [1, 2]), pt.rot90(r, [1, 2])
        assert np.array_equal(so[v], sr.numpy()) and np.array_equal(po[v], pr[0].numpy())
        assert np.array_equal(so[v + 1], sr.flip(1).numpy()) and np.array_equal(po[v + 1], pr.flip(1)[0].numpy())
        v += 2
