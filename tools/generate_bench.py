#!/usr/bin/env python3
"""End-to-end rate of the real producer: alphazero_amd/host/generate (512 games, 800 sims/move by default) streaming
8-symmetry rows to a local libzmq PULL socket that reads as fast as it can -- the reference trainer's side of
replay.hpp:120-150.  Prints generate's own closing JSON line plus what the sink counted.

    python tools/generate_bench.py --seconds 30 [--no-overlap] [--games 512] [--budget 800]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--budget", type=int, default=800)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--gpus", default=None)
    a = ap.parse_args()
    from test_wire import find_libzmq, free_port, ZMQ_PULL, ZMQ_RCVTIMEO
    z = find_libzmq()
    if z is None:
        raise SystemExit("no libzmq for the PULL side")
    z.zmq_ctx_new.restype = C.c_void_p
    z.zmq_socket.restype = C.c_void_p
    z.zmq_socket.argtypes = [C.c_void_p, C.c_int]
    z.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
    z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    z.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    z.zmq_close.argtypes = [C.c_void_p]
    port = free_port()
    exe = os.path.join(ROOT, "alphazero_amd", "host", "generate")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "ckpt_location.txt"), "w").write("gen_0.pt")
        env = dict(os.environ, AZ_GAMES=str(a.games), AZ_PORT=str(port), AZ_SECONDS=str(a.seconds))
        if a.no_overlap:
            env["AZ_NO_OVERLAP"] = "1"
        if a.gpus:
            env["AZ_GPUS"] = a.gpus
        fout = open(os.path.join(d, "stdout.txt"), "w")
        ferr = open(os.path.join(d, "stderr.txt"), "w")
        # stdin stays open for the whole run: the generator stops on "press enter" (generate.cpp:19-23) or end of input
        proc = subprocess.Popen([exe, "1", str(a.budget), "0"], cwd=d, env=env, stdin=subprocess.PIPE, stdout=fout, stderr=ferr)
        ctx = z.zmq_ctx_new()
        sock = z.zmq_socket(ctx, ZMQ_PULL)
        tmo = C.c_int(2000)
        z.zmq_setsockopt(sock, ZMQ_RCVTIMEO, C.byref(tmo), C.sizeof(tmo))
        assert z.zmq_connect(sock, b"tcp://127.0.0.1:%d" % port) == 0
        frames, nbytes, stop = [0], [0], [False]

        def sink():
            buf = C.create_string_buffer(1 << 16)
            while not stop[0]:
                n = z.zmq_recv(sock, buf, 1 << 16, 0)
                if n >= 0:
                    frames[0] += 1
                    nbytes[0] += n
        th = threading.Thread(target=sink)
        th.start()
        t0 = time.time()
        proc.wait(timeout=a.seconds + 240)
        proc.stdin.close()
        fout.close()
        ferr.close()
        out = open(os.path.join(d, "stdout.txt")).read()
        err = open(os.path.join(d, "stderr.txt")).read()
        time.sleep(2.5)                                   # let the sink read what is still in flight
        stop[0] = True
        th.join()
        z.zmq_close(sock)
    line = [ln for ln in out.splitlines() if ln.startswith('{"generate"')]
    if not line:
        sys.stderr.write(out[-2000:] + err[-2000:])
        raise SystemExit("generate printed no closing line (rc %d)" % proc.returncode)
    r = json.loads(line[-1])
    r["sink"] = {"messages": frames[0] // 3, "frames": frames[0], "megabytes": nbytes[0] / 1e6, "wall_s": time.time() - t0}
    print(json.dumps(r))


if __name__ == "__main__":
    main()
