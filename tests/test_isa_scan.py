"""No compiled kernel writes a data register of a three- or four-dword store too soon behind that store.

The MI355X reads such a store's data registers after the store has issued, and a write right behind it can get there
first: in az_conv3x3_t256s_kernel's compensated-mode epilogue v_pk_add_f32 one instruction behind
buffer_store_dwordx4 made the stored values differ from run to run (tools/probe/store_overwrite.hip shows the effect
in isolation).  The compiler pads the sequence (two wait states) only for stores without a scalar offset; the
kernels keep their own wait states there (wide_store_guard in tower.hip).  This test compiles every kernel source to
assembly with the Makefile's flags (no GPU needed) and runs tools/probe/wide_store_overwrite_scan.py over them."""
import concurrent.futures
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SOURCES = {"tower.hip": [], "tree.hip": ["-ffp-contract=off"], "blokus.hip": ["-ffp-contract=off"],
           "weights.hip": ["-ffp-contract=off"], "api.hip": []}


def _asm(src, flags, out):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "--cuda-device-only", "-S",
                        os.path.join(ROOT, "alphazero_amd", "csrc", src), "-o", out],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_write_into_a_wide_stores_data_registers_right_behind_it(tmp_path):
    with concurrent.futures.ThreadPoolExecutor(5) as pool:
        asms = list(pool.map(lambda kv: _asm(kv[0], kv[1], str(tmp_path / (kv[0] + ".s"))), SOURCES.items()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "wide_store_overwrite_scan.py"), *asms],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "az_conv3x3_t256s_kernel" in r.stdout and "too close" not in r.stdout      # (the scan lists the harmless sites too)


def test_the_scan_marks_what_it_should(tmp_path):
    """The two sequences that were seen to store other values, and the padded ones."""
    asm = tmp_path / "k.s"
    asm.write_text("k:\n"
                   "\tbuffer_store_dwordx4 v[10:13], v1, s[8:11], s5 offen\n\tv_mov_b32_e32 v10, 0\n"          # blind spot
                   "\tbuffer_store_dwordx4 v[20:23], v1, s[8:11], s5 offen\n\ts_nop 0\n\tv_pk_add_f32 v[20:21], v[20:21], v[2:3]\n"
                   "\tbuffer_store_dwordx4 v[30:33], v1, s[8:11], s5 offen\n\ts_nop 1\n\tv_pk_add_f32 v[30:31], v[30:31], v[2:3]\n"
                   "\tglobal_store_dwordx4 v[4:5], v[40:43], off\n\tv_add_u32_e32 v9, 1, v9\n\tv_mov_b32_e32 v40, 0\n"
                   "\tbuffer_store_dwordx4 v[50:53], v1, s[8:11], 0 offen\n\tv_add_u32_e32 v9, 1, v9\n\tv_mov_b32_e32 v50, 0\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "wide_store_overwrite_scan.py"), str(asm)],
                       capture_output=True, text=True, timeout=60)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 1 and len(lines) == 4 and any("no scalar offset, 1 wait" in ln and "x2" in ln for ln in lines), r.stdout
    marked = [ln for ln in lines if "too close" in ln]
    assert len(marked) == 3 and any("v_mov_b32" in ln and ", 0 wait" in ln for ln in marked) and \
        any("v_pk_add_f32" in ln and "1 wait" in ln for ln in marked) and any("no scalar offset, 1 wait" in ln for ln in marked)
