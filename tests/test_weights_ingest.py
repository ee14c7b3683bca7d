"""Checkpoint hand-off (util.h:60-139): a `torch::save(net, path)` archive
written by the reference's own PVNetwork (through the compiled harness) is
converted by tools/pt2blob.py into exactly the floats the harness dumped, and
the .azw container round-trips."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pt2blob  # noqa: E402

from alphazero_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "az_ref")


def test_azw_roundtrip(tmp_path):
    N, Cs = 5, [16, 16, 16]
    blob = synth.make_blob(N, 3, Cs, 9)
    path = str(tmp_path / "m_3.pt.azw")
    pt2blob.write_azw(path, N, 3, 1, Cs, blob)
    N2, ci, co, Cs2, b2 = pt2blob.read_azw(path)
    assert (N2, ci, co, Cs2) == (N, 3, 1, Cs) and np.array_equal(blob, b2)


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference harness not present")
def test_pt2blob_matches_reference_checkpoint(tmp_path):
    d = str(tmp_path)
    subprocess.run([REF, "mcts", "3", "3", "4", "1", "1", d, "5"], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=300)
    N, c_in, c_out, Cs, blob = pt2blob.state_to_blob(pt2blob.load_state(os.path.join(d, "model_0.pt")))
    ref = np.fromfile(os.path.join(d, "weights.bin"), np.float32)
    assert (N, c_in, c_out, Cs) == (3, 3, 1, [128] * 10)
    assert np.array_equal(blob, ref)
    assert blob.size == synth.blob_size(3, 3, [128] * 10)


def test_watch_converts_what_the_pointer_file_names(tmp_path):
    """pt2blob --watch: a checkpoint named by the pointer file gets its twin once; a missing checkpoint, an
    existing twin and an empty pointer file are left alone; relative paths resolve beside the pointer file."""
    ptr = tmp_path / "ckpt_location.txt"
    calls = []

    def fake_convert(path):
        calls.append(path)
        open(path + ".azw", "wb").write(b"x")
        return "ok"

    assert pt2blob.watch_once(str(ptr), fake_convert) is None          # no pointer file yet
    ptr.write_text("\n")
    assert pt2blob.watch_once(str(ptr), fake_convert) is None          # empty
    ptr.write_text("model_3.pt\n")
    assert pt2blob.watch_once(str(ptr), fake_convert) is None          # named, not written yet
    (tmp_path / "model_3.pt").write_bytes(b"pt")
    assert pt2blob.watch_once(str(ptr), fake_convert) == "ok"
    assert calls == [str(tmp_path / "model_3.pt")]
    assert pt2blob.watch_once(str(ptr), fake_convert) is None          # twin exists now
    assert len(calls) == 1
