"""The reference-named C++ shims (alphazero_amd/include/az/*.hpp) end to end
on the GPU, and this build's `generate` executable feeding a real libzmq PULL
socket in the reference trainer's format."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from test_wire import find_libzmq, free_port, ZMQ_PULL, ZMQ_RCVMORE, ZMQ_RCVTIMEO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "alphazero_amd", "host")


@pytest.mark.parametrize("nthreads", [1, 4])
def test_mcts_pvnetwork_shims_play_a_game(nthreads):
    """MCTS<Env>(nthreads, ...) through the shim: nthreads searcher threads become
    nthreads descents per round; the visit-count invariants hold either way."""
    exe = os.path.join(HOST, "shim_check")
    assert os.path.exists(exe), "build alphazero_amd/host first (__graft_entry__.build())"
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([exe, "50", str(nthreads)], cwd=d, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "shim ok" in out.stdout
        lines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("move ")]
        assert 5 <= len(lines) <= 9
        # first move: warm-up + search = 2*50 sims on a fresh root: sum = 100/101 (Q9, Q11)
        assert abs(float(lines[0][5]) - 100.0 / 101.0) < 1e-6
        for ln in lines:
            assert 0.9 < float(ln[5]) < 1.0
        assert os.path.exists(os.path.join(d, "shim_0.pt.azw"))     # created like util.h:94-97
        # the verbosity-3 report of the second move's search (mcts.hpp:545-575): every bucket is there and has time in
        # it, the counts are the search's (50 simulations, one evaluator call per round of the search)
        rep = {}
        for ln in out.stdout.splitlines():
            for key in ("(Root Prune)", "(Select)", "(Eval)", "(Step)", "(Net)", "(Append)", "(Backup)"):
                if ln.strip().startswith(key):
                    rep[key] = float(ln.split("Total:")[1].split("|")[0])
        assert set(rep) == {"(Root Prune)", "(Select)", "(Eval)", "(Step)", "(Net)", "(Append)", "(Backup)"}, out.stdout
        assert rep["(Select)"] > 0 and rep["(Step)"] > 0 and rep["(Net)"] > 0 and rep["(Append)"] > 0 and rep["(Backup)"] > 0
        assert abs(rep["(Eval)"] - (rep["(Step)"] + rep["(Net)"] + rep["(Append)"])) < 1e-3 * max(rep["(Eval)"], 1.0)
        assert "Simulation count: 50" in out.stdout and "Net count:" in out.stdout
        bl = [ln for ln in out.stdout.splitlines() if ln.startswith("blokus game")][0].split()
        assert 40 <= int(bl[2]) <= 84 and int(bl[-1]) == 5 * 400                # the Blokus Env shim played a whole game
        pv = [ln for ln in out.stdout.splitlines() if ln.startswith("pvnet")][0].split()
        assert abs(float(pv[2]) - 1.0) < 1e-4 and abs(float(pv[4])) < 1e-5


@pytest.mark.parametrize("contexts", [1, 2])
def test_checkpoint_swap_mid_run_goes_through_the_group_load_and_the_device_fold(contexts):
    """host/swap_check: ckpt_location.txt rewritten under flock while self-play runs; load() hands the new
    blob to az_group_load_weights (one host -> device copy, an ncclBroadcast when the group spans more than one
    GPU, BN fold + pack on the GPU) and the network that answers changes; a checkpoint that cannot be read (cut
    short) keeps the current weights."""
    exe = os.path.join(HOST, "swap_check")
    assert os.path.exists(exe), "build alphazero_amd/host first (__graft_entry__.build())"
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([exe, str(contexts)], cwd=d, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "swap ok: contexts %d" % contexts in out.stdout
        assert "cannot be read" in out.stdout and "Resuming with network model_1.pt" in out.stdout


@pytest.mark.parametrize("channels,fail_after,precision", [(None, None, None), ("32x4", None, None), (None, 2, None), (None, None, "fp16x3")])
@pytest.mark.skipif(find_libzmq() is None, reason="no libzmq for the PULL side")
def test_generate_executable_streams_rows_to_a_pull_socket(channels, fail_after, precision):
    """channels: AZ_NET_CHANNELS, a tower other than the compiled-in NetConfig 2 (a fresh checkpoint of that shape is
    written and read back like any other).  precision: AZ_PRECISION (the compensated mode runs lockstep rounds on the
    per-layer kernels: the start-up trial of the pair says so).  fail_after: the persistent pair loses its tree kernel at that launch
    (AZ_PAIR_FAIL_AFTER, a 50 ms time-out): the producer says so, restarts its games as lockstep rounds and keeps
    streaming."""
    exe = os.path.join(HOST, "generate")
    assert os.path.exists(exe)
    z = find_libzmq()
    z.zmq_ctx_new.restype = C.c_void_p
    z.zmq_socket.restype = C.c_void_p
    z.zmq_socket.argtypes = [C.c_void_p, C.c_int]
    z.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
    z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    z.zmq_getsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    z.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    z.zmq_close.argtypes = [C.c_void_p]
    port = free_port()
    N = 15
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "ckpt_location.txt"), "w").write("gen_0.pt")
        env = dict(os.environ, AZ_GAMES="32", AZ_PORT=str(port), AZ_SECONDS="60")
        if channels:
            env["AZ_NET_CHANNELS"] = channels
        if precision:
            env["AZ_PRECISION"] = precision
        if fail_after is not None:
            env.update(AZ_PAIR_FAIL_AFTER=str(fail_after), AZ_PAIR_TIMEOUT_MS="50")
        log = open(os.path.join(d, "generate.out"), "w")
        proc = subprocess.Popen([exe, "1", "8", "0"], cwd=d, env=env, stdin=subprocess.PIPE,
                                stdout=log, stderr=subprocess.PIPE)
        try:
            ctx = z.zmq_ctx_new()
            sock = z.zmq_socket(ctx, ZMQ_PULL)
            tmo = C.c_int(60000)
            z.zmq_setsockopt(sock, ZMQ_RCVTIMEO, C.byref(tmo), C.sizeof(tmo))
            assert z.zmq_connect(sock, b"tcp://127.0.0.1:%d" % port) == 0
            buf = C.create_string_buffer(1 << 16)
            got = 0
            for _ in range(64):                       # 8 positions x 8 symmetries
                sizes, mores = [], []
                for _f in range(3):
                    n = z.zmq_recv(sock, buf, 1 << 16, 0)
                    assert n >= 0, "no rows from generate"
                    more = C.c_int(0)
                    sz = C.c_size_t(C.sizeof(more))
                    z.zmq_getsockopt(sock, ZMQ_RCVMORE, C.byref(more), C.byref(sz))
                    sizes.append(n)
                    mores.append(bool(more.value))
                    if _f == 1:
                        pol = np.frombuffer(buf.raw[:n], np.float32)
                        assert 0.5 < pol.sum() <= 1.0 + 1e-5
                assert sizes == [3 * N * N, 4 * N * N, 8] and mores == [True, True, False]
                got += 1
            assert got == 64
            z.zmq_close(sock)
        finally:
            try:
                proc.stdin.write(b"\n")
                proc.stdin.close()
            except Exception:
                pass
            try:
                proc.wait(timeout=30)
            except subprocess.TimeoutExpired:
                proc.kill()
            log.close()
        out = open(os.path.join(d, "generate.out")).read()
        if fail_after is not None:
            assert "games restarted as lockstep rounds" in out, out[-2000:]
        elif precision == "fp16x3":
            assert "does not run here" in out and "using lockstep rounds" in out, out[-2000:]
        else:
            assert "lockstep rounds" not in out, out[-2000:]
