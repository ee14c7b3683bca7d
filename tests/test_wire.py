"""The generator->trainer wire format (replay.hpp:62-114) produced by the C++
ReplayBuffer(GEN) shim over the hand-written ZMTP 3.0 PUSH socket, received by a
REAL libzmq ZMQ_PULL socket exactly as the reference trainer does
(replay.hpp:120-150: three frames, RCVMORE set on the first two)."""
import ctypes as C
import glob
import os
import socket
import subprocess

import numpy as np
import pytest

import azo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SENDER = os.path.join(ROOT, "alphazero_amd", "host", "zmtp_send")
ZMQ_PULL, ZMQ_RCVMORE, ZMQ_RCVTIMEO = 7, 13, 27


def find_libzmq():
    for pat in ("/opt/conda/lib/libzmq.so.5", "/usr/lib/x86_64-linux-gnu/libzmq.so.5", "/usr/local/lib/libzmq.so*"):
        for p in glob.glob(pat):
            try:
                return C.CDLL(p)
            except OSError:
                pass
    return None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.skipif(find_libzmq() is None, reason="no libzmq to play the reference trainer's PULL side")
def test_push_stream_is_read_by_real_libzmq_pull():
    if not os.path.exists(SENDER):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(SENDER), "zmtp_send"])
    z = find_libzmq()
    z.zmq_ctx_new.restype = C.c_void_p
    z.zmq_socket.restype = C.c_void_p
    z.zmq_socket.argtypes = [C.c_void_p, C.c_int]
    z.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
    z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    z.zmq_getsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    z.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    z.zmq_close.argtypes = [C.c_void_p]
    z.zmq_ctx_term.argtypes = [C.c_void_p]
    N, rows = 3, 4
    port = free_port()
    proc = subprocess.Popen([SENDER, str(port), str(N), str(rows)], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    try:
        assert proc.stdout.readline().decode().strip() == "Path: tcp://127.0.0.1:%d" % port   # replay.hpp:49
        assert proc.stdout.readline().decode().strip() == "port %d" % port
        ctx = z.zmq_ctx_new()
        sock = z.zmq_socket(ctx, ZMQ_PULL)
        tmo = C.c_int(10000)
        z.zmq_setsockopt(sock, ZMQ_RCVTIMEO, C.byref(tmo), C.sizeof(tmo))
        assert z.zmq_connect(sock, b"tcp://127.0.0.1:%d" % port) == 0

        def recv_frame():
            buf = C.create_string_buffer(4096)
            n = z.zmq_recv(sock, buf, 4096, 0)
            assert n >= 0, "zmq_recv failed/timed out"
            more = C.c_int(0)
            sz = C.c_size_t(C.sizeof(more))
            z.zmq_getsockopt(sock, ZMQ_RCVMORE, C.byref(more), C.byref(sz))
            return buf.raw[:n], bool(more.value)

        for r in range(rows):
            s = np.array([(i * 7 + r) % 3 == 0 for i in range(3 * N * N)], np.uint8).reshape(3, N, N)
            p = np.array([(i + 1 + 100 * r) / 1000.0 for i in range(N * N)], np.float32).reshape(N, N)
            so, po = azo.replay_symmetries(N, s, p)
            for v in range(8):
                f1, m1 = recv_frame()
                f2, m2 = recv_frame()
                f3, m3 = recv_frame()
                assert (m1, m2, m3) == (True, True, False)                      # replay.hpp:133-146
                assert f1 == so[v].tobytes()
                assert f2 == po[v].tobytes()
                assert f3 == np.array([1.0, -1.0], np.float32).tobytes()
        z.zmq_close(sock)
        z.zmq_ctx_term(ctx)
    finally:
        try:
            proc.stdin.close()
        except Exception:
            pass
        proc.wait(timeout=10)
