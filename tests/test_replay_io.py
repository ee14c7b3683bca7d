"""The trainer's replay files (replay.hpp:163-208,241-274): files written by
libtorch's own torch::save (tests/golden/replay_cpp, produced by
`oracle/_ref/az_ref replayio write`) are read back, and files written by
alphazero_amd.replay_io are read by libtorch's torch::load when the compiled
reference harness is present."""
import os
import subprocess

import numpy as np
import pytest

from alphazero_amd import replay_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "replay_cpp")
REF = os.path.join(ROOT, "oracle", "_ref", "az_ref")


def expected():
    n, N = 5, 3
    s = (np.arange(n * 3 * N * N) % 3 == 0).astype(np.uint8).reshape(n, 3, N, N)
    p = (np.arange(n * N * N, dtype=np.float32) / np.float32(100.0)).reshape(n, 1, N, N)
    r = np.array([1, -1, -1, 1, 0, 0, 1, -1, 0, 0], np.float32).reshape(n, 2)
    return s, p, r


def test_reads_files_written_by_libtorch_cpp():
    s, p, r = replay_io.load_replay(GOLD)
    es, ep, er = expected()
    assert s.dtype == np.uint8 and np.array_equal(s, es)
    assert p.dtype == np.float32 and np.allclose(p, ep, atol=0, rtol=1e-7)
    assert np.array_equal(r, er)
    assert replay_io.load_replay(os.path.join(GOLD, "missing")) is None      # replay.hpp:173-177


def test_roundtrip_python(tmp_path):
    s, p, r = expected()
    replay_io.save_replay(str(tmp_path), s, p, r)
    s2, p2, r2 = replay_io.load_replay(str(tmp_path))
    assert np.array_equal(s, s2) and np.array_equal(p, p2) and np.array_equal(r, r2)


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference harness not present")
def test_libtorch_cpp_reads_python_written_files(tmp_path):
    s, p, r = expected()
    replay_io.save_replay(str(tmp_path), s, p, r)
    out = subprocess.run([REF, "replayio", "read", str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-400:]
    head, tail = out.stdout.strip().split("|")
    assert [int(x) for x in head.split()] == [5, 3, 1, 2]
    ssum, psum, rsum = tail.split()
    assert int(ssum) == int(s.sum()) and abs(float(psum) - float(p.sum())) < 1e-4 and abs(float(rsum) - 6.0) < 1e-6


def test_trainer_buffer_eviction_rules(tmp_path):
    """replay.hpp:158-159 vs 204-205: a received row evicts at size > max_size, a loaded row at
    size >= max_size; the threshold flag is raised after the append when receiving and before it
    when loading."""
    rows = [(np.full((3, 3, 3), i, np.uint8), np.full((1, 3, 3), i, np.float32), np.array([i, -i], np.float32))
            for i in range(7)]
    live = replay_io.TrainerBuffer(max_size=4, threshold=3)
    for i, r in enumerate(rows):
        live.receive(r)
        assert live.above_threshold == (i >= 2)                 # raised by the third row
    assert [int(r[0][0, 0, 0]) for r in live.rows] == [3, 4, 5, 6]              # keeps max_size rows
    loaded = replay_io.TrainerBuffer(max_size=4, threshold=3)
    loaded.load(rows[:3])
    assert not loaded.above_threshold                           # checked before each append: 0, 1, 2 != 3
    loaded.load(rows[3:])
    assert loaded.above_threshold
    assert [int(r[0][0, 0, 0]) for r in loaded.rows] == [4, 5, 6]               # keeps max_size - 1 rows
    live.save_dir(str(tmp_path))
    again = replay_io.TrainerBuffer(max_size=10, threshold=0)
    assert again.load_dir(str(tmp_path)) and len(again.rows) == 4
    assert np.array_equal(again.rows[0][2], np.array([3, -3], np.float32))
    assert not replay_io.TrainerBuffer().load_dir(str(tmp_path / "nothing"))
