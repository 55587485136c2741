import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import fluid_amd_loader
    return fluid_amd_loader.load()


@pytest.fixture(scope="session")
def oracle_path():
    path = os.path.join(ROOT, "oracle", "liboracle_fluid.so")
    if not os.path.exists(path):  # prebuilt on the GPU box; built on demand here
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    return path


@pytest.fixture(scope="session")
def oracle(pkg, oracle_path):
    """The CPU oracle bound through the same ctypes class as the product (checker only)."""
    return pkg.FluidLib(oracle_path)


@pytest.fixture(scope="session")
def hip(pkg):
    """The product library.  GPU tests fail loudly if it is missing: no fallback."""
    return pkg.FluidLib()
