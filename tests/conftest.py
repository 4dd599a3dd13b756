import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs on the CPU time this process can actually get: a GPU box shows every core of the machine in the affinity mask (256) while the
    # container's cgroup quota is 16 -- torch's default thread count then oversubscribes the quota sixteen-fold (bench.host_cpu_share)
    import torch
    from bench import host_cpu_share
    torch.set_num_threads(host_cpu_share()[0])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
