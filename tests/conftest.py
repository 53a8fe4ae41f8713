import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def load_golden(name):
    import json

    import numpy as np

    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    case = json.loads(str(z["case"]))
    return case, {k: z[k] for k in z.files if k != "case"}


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def has_gpu():
    import torch

    return torch.cuda.is_available()
