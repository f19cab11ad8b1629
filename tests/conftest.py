import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from hpgmg_testlib import Backend
    return Backend.oracle()


@pytest.fixture(scope="session")
def hip():
    from hpgmg_testlib import Backend
    return Backend.hip()
