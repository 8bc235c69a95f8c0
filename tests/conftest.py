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
    config.addinivalue_line("markers", "slow: full-size oracle comparisons (tens of seconds of CPU oracle time each)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def sd_from_npz(npz, prefix):
    """Inverse of make_golden.sd_numpy: keys 'prefix/a/b' -> 'a.b' torch tensors."""
    import torch
    out = {}
    for k in npz.files:
        if k.startswith(prefix):
            out[k[len(prefix):].replace("/", ".")] = torch.from_numpy(np.asarray(npz[k]))
    return out
