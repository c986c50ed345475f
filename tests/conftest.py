import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """(arrays, expected) of one golden fixture; arrays is {} when the fixture
    has no .npz part."""
    arrays = {}
    npz = os.path.join(GOLDEN, name + ".npz")
    if os.path.exists(npz):
        with np.load(npz, allow_pickle=False) as z:
            arrays = {k: z[k] for k in z.files}
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        expected = json.load(f)
    return arrays, expected


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
