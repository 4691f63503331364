"""CPU oracle of the binarised-graph statistics the reference evaluates samples with (TEST INFRASTRUCTURE ONLY).

Parity status: PINNED against `eval_torch_batch` of the reference (evaluation/stats.py:382-393, methods degree /
cluster / spectral) by tests/golden/make_golden_mmd.py -> tests/golden/golden_mmd.npz.  The orbit statistic (ORCA
subprocess, evaluation/stats.py:200-277) and NSPDK (EDeN, absent here) are not restated.

Plain numpy on adjacency matrices; no networkx.  Citations are file:line under /root/reference.
"""
import numpy as np


def clean_adj(adj: np.ndarray) -> np.ndarray:
    """adjs_to_graphs, evaluation/stats.py:291-305: binary symmetric adjacency -> graph without self loops and
    without isolated nodes (a single node when nothing is left)."""
    a = (np.asarray(adj) != 0).astype(np.float64)
    np.fill_diagonal(a, 0.0)
    a = np.maximum(a, a.T)
    keep = a.sum(1) > 0
    if not keep.any():
        return np.zeros((1, 1))
    return a[np.ix_(keep, keep)]


def degree_hist(a: np.ndarray) -> np.ndarray:
    """nx.degree_histogram, evaluation/stats.py:26-30."""
    return np.bincount(a.sum(1).astype(np.int64)).astype(np.float64)


def clustering_hist(a: np.ndarray, bins: int = 100) -> np.ndarray:
    """nx.clustering values -> histogram over [0, 1], evaluation/stats.py:72-80."""
    d = a.sum(1)
    tri = np.einsum("ij,jk,ki->i", a, a, a) / 2.0
    denom = d * (d - 1) / 2.0
    c = np.where(denom > 0, tri / np.maximum(denom, 1e-300), 0.0)
    hist, _ = np.histogram(c, bins=bins, range=(0.0, 1.0), density=False)
    return hist.astype(np.float64)


def spectral_pmf(a: np.ndarray) -> np.ndarray:
    """Eigenvalues of the normalised Laplacian -> 200-bin pmf, evaluation/stats.py:128-135."""
    d = a.sum(1)
    inv = np.where(d > 0, 1.0 / np.sqrt(np.maximum(d, 1e-300)), 0.0)
    lap = np.eye(len(a)) * (d > 0) - inv[:, None] * a * inv[None, :]
    eigs = np.linalg.eigvalsh(lap)
    pmf, _ = np.histogram(eigs, bins=200, range=(-1e-5, 2), density=False)
    return pmf / pmf.sum()


def gaussian_tv(x, y, sigma=1.0):
    """evaluation/mmd.py:82-95 (zero padding to equal length, :19-31)."""
    n = max(len(x), len(y))
    xx, yy = np.zeros(n), np.zeros(n)
    xx[:len(x)], yy[:len(y)] = x, y
    dist = np.abs(xx - yy).sum() / 2.0
    return np.exp(-dist * dist / (2 * sigma * sigma))


def compute_mmd(s1, s2, sigma=1.0):
    """evaluation/mmd.py:115-163 with is_hist=True and the Gaussian-TV kernel."""
    s1 = [s / s.sum() if s.sum() != 0 else s for s in s1]
    s2 = [s / s.sum() if s.sum() != 0 else s for s in s2]

    def disc(a, b):
        return sum(gaussian_tv(x, y, sigma) for x in a for y in b) / (len(a) * len(b))

    return disc(s1, s1) + disc(s2, s2) - 2 * disc(s1, s2)


def eval_adjs(ref_adjs, pred_adjs):
    """eval_torch_batch(ref, pred, methods=['degree', 'cluster', 'spectral']), evaluation/stats.py:362-393."""
    ref = [clean_adj(a) for a in ref_adjs]
    pred = [clean_adj(a) for a in pred_adjs]
    out = {
        "degree": compute_mmd([degree_hist(a) for a in ref], [degree_hist(a) for a in pred]),              # :63
        "cluster": compute_mmd([clustering_hist(a) for a in ref], [clustering_hist(a) for a in pred], 0.1),  # :119
        "spectral": compute_mmd([spectral_pmf(a) for a in ref], [spectral_pmf(a) for a in pred]),          # :170
    }
    out["average"] = float(np.mean(list(out.values())))                                                     # :377
    return out
