#!/usr/bin/env python3
"""Lists, per kernel of a gfx950 assembly file (hipcc --cuda-device-only -S), the VALU instructions that write a data
register of a three- or four-dword store a few instructions behind it, by opcode and by the wait states in between
(an instruction counts one, s_nop N counts N + 1).

Why: the MI355X reads a wide store's data registers after the store has issued, a quarter wave at a time, and an
instruction behind the store that writes one of them can get there first (tools/probe/store_overwrite.hip, 1 M lanes a
run).  Lanes that stored other values than they held:
  buffer_store_dwordx4 with a scalar (SGPR) offset: writer right behind: v_pk_add_f32 23 %, plain VALU up to 0.09 %;
      one wait state in between: none.
  global_store_dwordx4 (and buffer stores without a scalar offset): writer right behind: 23-47 %; one wait state in
      between: v_pk_add_f32 23 %, v_mov_b32 up to 0.05 %; two wait states: none.
  two-dword stores: none.
The compiler pads such a sequence (s_nop 1, two wait states, on this target) only when the store has no scalar offset.  Marked, and exit
status 1: a writer with no wait state in between; a writer with fewer than two behind a store without a scalar
offset; a packed fp32 writer with fewer than two behind any wide store.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S alphazero_amd/csrc/tower.hip -o /tmp/tower.s
    python tools/probe/wide_store_overwrite_scan.py /tmp/tower.s [--within 4]"""
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
            buf = m.group(1).startswith("buffer")
            data = regs(ops[0]) if buf else regs(ops[1])
            sgpr_off = buf and len(ops) > 3 and ops[3].split()[0].startswith("s")
            k, j, waits = 0, i + 1, 0
            while k < a.within and j < len(lines):
                t = lines[j].strip()
                j += 1
                if not t or t[0] in ";.":
                    continue
                k += 1
                w = re.match(r"^(v_\w+)\s+(\S+?),", t)
                if w and not w.group(1).startswith("v_cmp") and regs(w.group(2)) & data:
                    packed = w.group(1).startswith("v_pk_") and w.group(1).endswith("f32")
                    flag = waits == 0 or (not sgpr_off and waits < 2) or (packed and waits < 2)
                    tally[(kern, waits, w.group(1), "scalar offset" if sgpr_off else "no scalar offset", flag)] += 1
                n = re.match(r"^s_nop\s+(\d+)", t)
                waits += int(n.group(1)) + 1 if n else 1
        for (kern, waits, op, off, flag), n in sorted(tally.items()):
            bad += 1 if flag else 0
            print("%s  %s  %s, %d wait state(s) in between  %s  x%d%s" % (path, kern[:64], off, waits, op, n, "   <-- too close" if flag else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
