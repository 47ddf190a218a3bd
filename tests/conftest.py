import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_budget():
    """CPU threads this process may really use: the cgroup quota where there is one, never more than 16.  On the GPU boxes
    os.cpu_count() says 256 while the job's share is 16: torch then runs the oracle's CPU kernels on 128 threads and the at-size
    tests take EIGHT times as long (one 150k-voxel oracle test: 52 s against 6.5 s with 16 threads; round 4)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, quota // int(g.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(16, n))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    n = _cpu_budget()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))            # the rank workers and the C++ checker inherit it
    try:
        import torch
        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    except Exception:                                           # noqa: BLE001  (a test that needs torch will say so itself)
        pass


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
