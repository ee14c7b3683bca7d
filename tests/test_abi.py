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


def test_host_rules_match_oracle_on_random_playouts():
    import azo
    rng = np.random.default_rng(3)
    for N, K in ((3, 3), (7, 4), (9, 5), (15, 5)):
        for _ in range(20):
            ca, ta = azo.env_reset(N)
            cb, tb = ca.copy(), 0
            while True:
                a = int(rng.choice(azo.env_legal(N, ca)))
                ca, ta, ra, da = azo.env_step(N, K, ca, ta, a)
                cb, tb, rb, db = capi.host_env_step(N, K, cb, tb, a)
                assert np.array_equal(ca, cb) and ta == tb and np.array_equal(ra, rb) and da == db
                if da:
                    break
            # an occupied cell is rejected by both
            assert capi.host_env_step(N, K, cb, tb, a)[3] < 0 and azo.env_step(N, K, ca, ta, a)[3] < 0


def test_ctypes_structures_match_the_header_layout(tmp_path):
    """alphazero_amd/capi.py restates az_config and az_stats by hand: compile a probe against
    include/az_hip.h with the C compiler and compare sizes and every field offset."""
    import ctypes
    import subprocess
    fields = {"az_config": [f for f, _ in capi.AzConfig._fields_], "az_stats": [f for f, _ in capi.AzStats._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "az_hip.h"', 'int main(void) {']
    for st, fs in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    src.append('return 0; }')
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, cls in (("az_config", capi.AzConfig), ("az_stats", capi.AzStats)):
        assert int(got[st]) == ctypes.sizeof(cls), st
        for f in fields[st]:
            assert int(got["%s.%s" % (st, f)]) == getattr(cls, f).offset, (st, f)


def test_null_context_is_refused_without_a_gpu():
    """Argument checks come before any HIP call."""
    lib = capi.load()
    assert lib.az_sync(None) == capi.AZ_E_ARG
    assert lib.az_selfplay_run(None, 1) == capi.AZ_E_ARG
    assert lib.az_weights_size(None) == 0
    assert lib.az_ctx_create(None, None) == capi.AZ_E_ARG
    assert b"null argument" in lib.az_last_error()
    assert lib.az_ctx_destroy(None) == capi.AZ_OK
