"""The compiled tower kernels hold no packed-fp32 write into the data registers of a three- or four-dword store right
behind that store: with v_pk_add_f32 one instruction behind buffer_store_dwordx4 the stored values differed from run
to run on the MI355X (az_conv3x3_t256s_kernel's compensated-mode epilogue, which now keeps wait states there), and
the compiler does not list the case for stores with a scalar offset.  Compiles tower.hip to assembly with the
Makefile's flags (no GPU needed) and runs tools/probe/wide_store_overwrite_scan.py over it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_packed_fp32_write_right_behind_a_wide_store(tmp_path):
    asm = str(tmp_path / "tower.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "alphazero_amd", "csrc", "tower.hip"), "-o", asm],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "wide_store_overwrite_scan.py"), asm],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "az_conv3x3_t256s_kernel" in r.stdout          # the scan saw the kernels (it lists the harmless sites too)
