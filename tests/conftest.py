import os
import sys


def _usable_cpus():
    """CPU threads this container may really use: the cgroup quota / affinity mask, not the 256 cores a GPU box reports."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


# Before ANY OpenMP runtime is loaded (numpy, torch, the oracle, the host layer): a team sized to the quota and no
# spin-waiting.  Round 1's driver run died at its 1200 s limit with 256 spinning libgomp threads on a 16-CPU quota.
os.environ.setdefault("OMP_NUM_THREADS", str(min(_usable_cpus(), 16)))
if _usable_cpus() < (os.cpu_count() or 1):          # a quota below the visible cores (the GPU box): spinning threads starve each other
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OMP_PROC_BIND", "false")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import pytest  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "modes: solver modes SURVEY.md section 2 marks OUT OF SCOPE (MGPCG, CG bottom solver, U-/V-cycle shapes, MGSolve of this host "
                                       "layer); deselected unless the -m expression names them: -m 'gpu and modes'")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` is the hot path's suite; the out-of-scope solver modes only run when asked for by name."""
    if "modes" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("modes") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def oracle():
    from hpgmg_testlib import Backend
    return Backend.oracle()


@pytest.fixture(scope="session")
def hip():
    from hpgmg_testlib import Backend
    return Backend.hip()
