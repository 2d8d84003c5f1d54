import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracles run thousands of tiny matmuls (the reward restatement: 4 (d - 1) M encoder calls): torch's default of one
    # thread per host core (128 on a GPU box whose job owns a 16-core share) oversubscribes and has been seen to take minutes
    torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_params(g, prefix="param."):
    """12 trainable tensors (prior_mean / prior_std are frozen constants, dropped)."""
    out = {}
    for k, v in g.items():
        if k.startswith(prefix) and "prior" not in k:
            out[k[len(prefix):]] = torch.from_numpy(v.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
