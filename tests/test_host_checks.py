"""Host-side logic that needs no GPU: the generator's AZ_GPUS parser, the per-GPU worker thread of the group API and
az_group_create's argument / missing-RCCL error paths (alphazero_amd/host/cpu_check.cpp against libazhip.so), and the
sanitizer builds of that program and of the oracle (AddressSanitizer + UBSan, ThreadSanitizer)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "alphazero_amd", "host")


def test_cpu_check_parse_gpus_workers_and_group_error_paths():
    exe = os.path.join(HOST, "cpu_check")
    assert os.path.exists(exe), "build alphazero_amd/host first (__graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cpu_check ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")
def test_host_logic_and_oracle_are_clean_under_the_sanitizers():
    for d in (HOST, os.path.join(ROOT, "oracle")):
        out = subprocess.run(["make", "-s", "-C", d, "sanitize"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "runtime error" not in out.stderr and "Sanitizer" not in out.stderr, out.stderr
