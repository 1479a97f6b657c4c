"""pytest configuration: the ``gpu`` marker, import paths and golden-fixture helpers."""
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))   # the ``rnampnn`` drop-in package
sys.path.insert(0, REPO)                                  # ``oracle`` (test infrastructure)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))
FULL_CASES = ["c1_1b23_k16_P66", "phantom_n5_T8_k6", "ragged_k30", "alt_cfg_k4"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (arrays dict, hparams dict, state-dict shapes dict) of one committed fixture."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    arrs = {k: z[k] for k in z.files if k not in ("hparams", "state_keys")}
    hp = json.loads(bytes(z["hparams"]).decode())
    shapes = {k: tuple(v) for k, v in json.loads(bytes(z["state_keys"]).decode()).items()}
    return arrs, hp, shapes


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
