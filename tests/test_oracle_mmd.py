"""Pin oracle/graph_stats.py (degree / clustering / spectral Gaussian-TV MMD on binarised graphs) against the values
the reference's evaluation code returned for the same graph sets (tests/golden/make_golden_mmd.py)."""
import os

import numpy as np

from oracle import graph_stats as GS
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mmd_statistics_match_reference(golden):
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_mmd.npz"), allow_pickle=False) as z:
        ref = {k: z[k] for k in z.files}
    for key, (a, b) in H.mmd_graph_sets(golden).items():
        r = GS.eval_adjs(a, b)
        mine = np.array([r["degree"], r["cluster"], r["spectral"], r["average"]])
        assert np.allclose(mine, ref[f"mmd/{key}"], rtol=1e-9, atol=1e-12), (key, mine, ref[f"mmd/{key}"])


def test_edge_cases():
    """Empty graph -> single node (evaluation/stats.py:302-303); identical sets -> MMD 0."""
    empty = np.zeros((5, 5))
    assert GS.clean_adj(empty).shape == (1, 1)
    a = H.synthetic_graphs("e", 4, 12, 5, 12, 0.4, 1)
    r = GS.eval_adjs(a, a)
    assert abs(r["degree"]) < 1e-12 and abs(r["cluster"]) < 1e-12 and abs(r["spectral"]) < 1e-12
    r2 = GS.eval_adjs(a, np.concatenate([a[:3], empty[None, :12 - 7, :12 - 7].repeat(1, 0) if False else np.zeros((1, 12, 12))]))
    assert np.isfinite(list(r2.values())).all()
