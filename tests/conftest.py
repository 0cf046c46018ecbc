import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The parity tests decode EVERY stream shape, lone members and single frames included, so they run with the bid policy
# switched off; the policy itself (the product's default, LA_GPU_BID=auto) is what tests/test_gpu_bid_policy.py checks.
os.environ.setdefault("LA_GPU_BID", "all")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side native pieces (oracle, host C, generators) are cheap to (re)build with gcc.
    for d in ("oracle", "libarchive_amd/host", "tools"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import libarchive_amd as la
    ctx = la.GpuContext(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    yield ctx
    ctx.close()
