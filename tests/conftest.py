import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    from clap_amd import _lib
    _lib.check(_lib.lib().clapgpu_init(0), "clapgpu_init")
    return "cuda:0"


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="session")
def process_group(cuda_device):
    """World-size-1 RCCL process group, one per test session (GPU tests run in one process: an RCCL communicator is
    not something to create and tear down per module)."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(cuda_device))
    yield dist
    dist.destroy_process_group()


def pytest_sessionfinish(session, exitstatus):
    """Worst per-object relative errors the floating-point parity tests saw (tests/helpers.py), for profiles/."""
    try:
        from helpers import PARITY_BOUNDS
    except Exception:
        return
    if not PARITY_BOUNDS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_bounds.json"), "w") as f:
            json.dump({"bar": "equality of values (round 4); north_star's bar is 1e-5 relative",
                       "norm": "per quantity: the worst max|got-ref| over one object / max|ref| over the same object (0.0 = every object "
                               "equal) and the number of objects compared",
                       "worst": dict(sorted(PARITY_BOUNDS.items()))}, f, indent=1)
    except OSError:
        pass
