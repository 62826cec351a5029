import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _granted_cores():
    """cores this process may really use: the affinity mask cut to the cgroup's CPU-time quota (a GPU box shows all 256 cores of its host
    and grants one GPU's share of them: PyTorch-CPU with a thread per visible core -- the oracle's side of the parity tests -- then
    runs several times slower than with a thread per granted core)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    import torch
    torch.set_num_threads(_granted_cores())
    yield

