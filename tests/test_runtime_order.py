"""One HIP runtime per process, whichever of alphazero_amd.capi and torch is imported first."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CPU_SNIPPET = """
import sys
sys.path.insert(0, %r)
from alphazero_amd import capi
capi.load()
first = capi.hip_runtimes_mapped()
import torch
both = capi.hip_runtimes_mapped()
assert len(first) == 1 and both == first, (first, both)
assert os.path.dirname(first[0]) == os.path.join(os.path.dirname(torch.__file__), "lib"), first
print("one runtime:", first[0])
"""

GPU_SNIPPET = """
import os, sys
sys.path.insert(0, %r)
from alphazero_amd import capi                     # BEFORE torch, and the GPU is touched before torch too
eng = capi.Engine(5, 4, 2, eval_mode=capi.AZ_EVAL_HASH, max_budget=8)
eng.env_reset()
out = eng.search(8)
import torch
import torch.distributed as dist
assert len(capi.hip_runtimes_mapped()) == 1, capi.hip_runtimes_mapped()
assert torch.cuda.is_available()
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = "29741"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
assert float(t.sum()) == 4.0
# a torch allocation is usable by the library (same runtime, same address space)
blob = torch.zeros(16, device="cuda")
dist.destroy_process_group()
print("nccl after capi ok", out.shape)
"""


def _run(snippet):
    r = subprocess.run([sys.executable, "-c", "import os\n" + snippet % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_capi_before_torch_maps_torchs_runtime_only():
    assert "one runtime:" in _run(CPU_SNIPPET)


@pytest.mark.gpu
def test_capi_first_then_torch_nccl_on_one_rank():
    assert "nccl after capi ok" in _run(GPU_SNIPPET)
