import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(REPO, "tests", "data", "Set5")
ASSETS = os.path.join(REPO, "lerf-pytorch_amd", "assets", "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import lerf_oracle
    return lerf_oracle


@pytest.fixture(scope="session")
def luts_g(oracle):
    return oracle.load_luts(os.path.join(ASSETS, "lerf-g"), linear=False)


@pytest.fixture(scope="session")
def luts_l(oracle):
    return oracle.load_luts(os.path.join(ASSETS, "lerf-l"), linear=True)
