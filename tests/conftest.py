import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import helpers as H
    if not os.path.exists(H.oracle_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(H.ROOT, "oracle"), "liboracle.so"])
    return H.oracle()
