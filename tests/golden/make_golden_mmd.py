"""Golden MMD statistics from the REFERENCE's evaluation code (evaluation/stats.py eval_torch_batch, methods degree /
cluster / spectral) on fixed binarised graph sets.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mmd.py        (build container only)

Import shims (no arithmetic of the three statistics goes through them): `pyemd` and `eden.graph` are absent here and
only serve the EMD / NSPDK kernels; networkx 3.x renamed `from_numpy_matrix` to `from_numpy_array` (the reference
pins 2.8.7, setup/requirements.txt:9).
Graph sets: (a) thresholded final samples the reference's own EDMSampler produced (tests/golden/golden.npz) against
synthetic "test set" graphs; (b) two synthetic sets of community-like random graphs.
"""
import os
import sys
import types

import numpy as np
import networkx as nx

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
for name in ("pyemd", "eden", "eden.graph"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["eden.graph"].vectorize = None
sys.modules["pyemd"].emd = None
if not hasattr(nx, "from_numpy_matrix"):
    nx.from_numpy_matrix = nx.from_numpy_array
sys.path.insert(0, "/root/reference")

import torch                                                  # noqa: E402
from evaluation import stats as RS                            # noqa: E402  (reference)

from tests.helpers import mmd_graph_sets                      # noqa: E402

RS.PRINT_TIME = False
OUT = {}
gold = np.load(os.path.join(HERE, "golden.npz"))
for key, (ref, pred) in mmd_graph_sets(gold).items():
    res = RS.eval_torch_batch(torch.from_numpy(ref), torch.from_numpy(pred), methods=["degree", "cluster", "spectral"])
    OUT[f"mmd/{key}"] = np.array([res["degree"], res["cluster"], res["spectral"], res["average"]], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "golden_mmd.npz"), **OUT)
print({k: v.round(6).tolist() for k, v in OUT.items()})
