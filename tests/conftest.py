import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def golden_params(gold, prefix):
    pre = prefix + "_param|"
    return {k[len(pre):].replace("|", "/"): v for k, v in gold.items() if k.startswith(pre)}


@pytest.fixture(scope="session")
def golden_estimators():
    return load_golden("estimators_oracle_driven.npz")


@pytest.fixture(scope="session")
def golden_j1j2():
    return load_golden("j1j2_matrix_elements.npz")


@pytest.fixture(scope="session")
def golden_product():
    return load_golden("ising_product_state.npz")


def all_configs(N):
    """(2**N, N) int32, row k = binary digits of k, site 0 most significant."""
    return ((np.arange(2 ** N)[:, None] >> np.arange(N)[::-1]) & 1).astype(np.int32)
