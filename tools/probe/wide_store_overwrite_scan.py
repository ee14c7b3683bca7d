#!/usr/bin/env python3
"""Lists, per kernel of a gfx950 assembly file (hipcc --cuda-device-only -S), the VALU instructions that write a data
register of a three- or four-dword store within a few instructions behind it, by opcode and distance.

Why: in az_conv3x3_t256s_kernel's compensated-mode epilogue, v_pk_add_f32 one instruction behind a
buffer_store_dwordx4 with a scalar offset, writing that store's data registers, made the stored values differ from
run to run on the MI355X; the compiler's hazard table covers such a store only without a scalar offset.  The same
distance with v_accvgpr_read_b32, v_and_b32, v_lshlrev_b32 or v_med3_f32 as the writer occurs in the kernels whose
outputs the tests hold bit for bit, so only packed fp32 writers are marked here.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S alphazero_amd/csrc/tower.hip -o /tmp/tower.s
    python tools/probe/wide_store_overwrite_scan.py /tmp/tower.s [--within 4]
Exit status 1 if a packed fp32 writer sits within two instructions of such a store."""
import argparse
import collections
import re
import sys

STORE = re.compile(r"^\s*((?:buffer|global|flat)_store_dwordx[34])\s+(.*)$")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="+")
    ap.add_argument("--within", type=int, default=4)
    a = ap.parse_args()
    bad = 0
    for path in a.asm:
        lines = open(path).read().splitlines()
        kern, tally = "?", collections.Counter()
        for i, ln in enumerate(lines):
            if ln and not ln[0].isspace() and not ln.startswith(".") and ln.rstrip().split(";")[0].rstrip().endswith(":"):
                kern = ln.split(":")[0]
            m = STORE.match(ln)
            if not m:
                continue
            ops = [o.strip() for o in m.group(2).split(",")]
            data = regs(ops[0]) if m.group(1).startswith("buffer") else regs(ops[1])
            k, j = 0, i + 1
            while k < a.within and j < len(lines):
                t = lines[j].strip()
                j += 1
                if not t or t[0] in ";.":
                    continue
                k += 1
                w = re.match(r"^(v_\w+)\s+(\S+?),", t)
                if w and not w.group(1).startswith("v_cmp") and regs(w.group(2)) & data:
                    tally[(kern, k, w.group(1))] += 1
        for (kern, k, op), n in sorted(tally.items()):
            mark = ""
            if op.startswith("v_pk_") and op.endswith("f32") and k <= 2:
                mark, bad = "   <-- packed fp32 writer right behind the store", bad + 1
            print("%s  %s  distance %d  %s  x%d%s" % (path, kern[:64], k, op, n, mark))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
