"""Checkpoint hand-off (util.h:60-139): a `torch::save(net, path)` archive
written by the reference's own PVNetwork (through the compiled harness) is
converted by tools/pt2blob.py -- and by the generator's own C++ reader
(az/ptarchive.hpp, no libtorch) -- into exactly the floats the harness dumped,
and the .azw container round-trips."""
import os
import struct
import subprocess
import sys
import zipfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pt2blob  # noqa: E402

from alphazero_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "az_ref")


def _reference_checkpoint(d):
    """model_0.pt (torch::save of the reference's PVNetwork) and weights.bin (its floats) in d.  The harness
    runs the reference's own searcher and evaluator threads (mcts.hpp:281-282 signals evaluator.hpp:88 and then
    waits for the answer); a run normally takes under a second, one was seen not to finish in five minutes, so a
    run gets a minute and up to three starts."""
    for attempt in range(3):
        try:
            subprocess.run([REF, "mcts", "3", "3", "4", "1", "1", d, "5"], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=60)
            return
        except subprocess.TimeoutExpired:
            if attempt == 2:
                raise


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
    _reference_checkpoint(d)
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


PT2AZW = os.path.join(ROOT, "alphazero_amd", "host", "pt2azw")


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference harness not present")
def test_cpp_reader_matches_reference_checkpoint(tmp_path):
    """The generator's own reader (zip directory + pickle-as-data + raw storages) on the file the reference's
    trainer would publish: the blob is the harness's dump bit for bit, the shape is read off the tensors."""
    d = str(tmp_path)
    _reference_checkpoint(d)
    out = subprocess.run([PT2AZW, os.path.join(d, "model_0.pt"), os.path.join(d, "cpp.azw")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    N, c_in, c_out, Cs, blob = pt2blob.read_azw(os.path.join(d, "cpp.azw"))
    assert (N, c_in, c_out, Cs) == (3, 3, 1, [128] * 10)
    assert np.array_equal(blob, np.fromfile(os.path.join(d, "weights.bin"), np.float32))


def _archive(path, pkl, storages=()):
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr("m/data.pkl", pkl)
        for k, raw in storages:
            z.writestr("m/data/%s" % k, raw)


def test_cpp_reader_treats_the_pickle_as_data_and_refuses_what_it_does_not_know(tmp_path):
    """Nothing in a checkpoint is executed: a REDUCE of os.system is recorded as a value and ignored, an opcode
    outside the small set such archives use ends the read, and files that are no archive say so."""
    def run(path):
        return subprocess.run([PT2AZW, str(path), str(tmp_path / "out.azw")], capture_output=True, text=True, timeout=60)

    marker = tmp_path / "ran"
    evil = b"\x80\x02cos\nsystem\nX" + struct.pack("<I", len("touch %s" % marker)) + ("touch %s" % marker).encode() + b"\x85R."
    _archive(tmp_path / "evil.pt", evil)
    out = run(tmp_path / "evil.pt")
    assert out.returncode == 1 and "no tensor CBlock.conv.weight" in out.stderr
    assert not marker.exists()

    _archive(tmp_path / "inst.pt", b"\x80\x02(ios\nsystem\n.")            # INST: not an opcode a checkpoint uses
    out = run(tmp_path / "inst.pt")
    assert out.returncode == 1 and "opcode 0x69" in out.stderr

    (tmp_path / "short.pt").write_bytes(b"PK\x03\x04 cut short")
    out = run(tmp_path / "short.pt")
    assert out.returncode == 1 and "not a zip archive" in out.stderr

    # a tensor whose storage is shorter than the pickle claims
    t = (b"\x80\x02c__torch__\nModule\n)\x81}(X\x06\x00\x00\x00CBlockc__torch__\nModule\n)\x81}(X\x04\x00\x00\x00convc__torch__\nModule\n)\x81}("
         b"X\x06\x00\x00\x00weightctorch._utils\n_rebuild_tensor_v2\n((X\x07\x00\x00\x00storagectorch\nFloatStorage\nX\x01\x00\x00\x000X\x03\x00\x00\x00cpuK\x08tQK\x00(K\x02K\x04t(K\x04K\x01t\x89NtR"
         b"ububub.")
    _archive(tmp_path / "trunc.pt", t, [("0", b"\0" * 8)])
    out = run(tmp_path / "trunc.pt")
    assert out.returncode == 1 and "shorter than its pickle says" in out.stderr
