import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    if pkg.lib().mods_device_count() <= 0:
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    ctx = pkg.Context(0, 1920, 1080, 2)
    yield ctx
    ctx.close()
