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


def load_golden(name):
    """npz -> dict of torch tensors (float16-stored weights are widened back to float32)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype == np.float16:
            a = a.astype(np.float32)
        if a.dtype.kind in "US":  # arrays of names (e.g. state-dict keys) stay Python strings
            out[k] = [str(x) for x in a.reshape(-1)]
            continue
        out[k] = torch.from_numpy(a) if a.shape != () else torch.tensor(a.item())
    return out


def sub(d, prefix):
    """Entries of d under 'prefix' with the prefix stripped."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


@pytest.fixture(autouse=True)
def _reset_functional_state():
    """The process-wide switches of speechain_amd.functional go back to their defaults after every test, also after a failing one."""
    yield
    fn = sys.modules.get("speechain_amd.functional")
    if fn is not None:
        fn.set_compute_dtype(None)
        fn.set_attention_probs(True)
