import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Golden vectors produced by the reference modules (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "golden.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_meta():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_train():
    """Train-step goldens (loss + gradient fingerprints) from the reference modules (tests/golden/make_golden_train.py)."""
    import json
    path = os.path.join(ROOT, "tests", "golden", "golden_train.npz")
    with np.load(path, allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    with open(os.path.join(ROOT, "tests", "golden", "golden_train_meta.json")) as f:
        return arrays, json.load(f)


@pytest.fixture(scope="session")
def golden_nodeadj():
    """Molecule-path goldens (NodeAdjSwinGNN / NodeAdjPrecond / NodeAdjEDMSampler) from the reference modules
    (tests/golden/make_golden_nodeadj.py)."""
    import json
    path = os.path.join(ROOT, "tests", "golden", "golden_nodeadj.npz")
    with np.load(path, allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    with open(os.path.join(ROOT, "tests", "golden", "golden_nodeadj_meta.json")) as f:
        return arrays, json.load(f)


@pytest.fixture(scope="session")
def golden_nodeadj_train():
    """Molecule-path train-step goldens from the reference modules (tests/golden/make_golden_nodeadj_train.py)."""
    path = os.path.join(ROOT, "tests", "golden", "golden_nodeadj_train.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
