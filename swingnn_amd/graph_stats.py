"""Binarised-graph statistics of the reference's evaluation on the device: degree / clustering /
spectral histograms and the Gaussian-TV MMD of evaluation/stats.py:26-175, evaluation/mmd.py:82-163, for graphs given
as adjacency tensors [B,N,N] on an MI355X (what EDMSampler.sample + io.quantise_samples produce).

    eval_adjs(ref_adjs, pred_adjs) -> {'degree', 'cluster', 'spectral', 'average'}      # eval_torch_batch, :362-393

The histograms, the Laplacian, the numpy-compatible binning and the MMD are HIP kernels (csrc/graph_stats.hip); the
eigenvalues of the normalised Laplacian come from torch.linalg.eigvalsh (rocSOLVER, fp64) -- a library solver, like
RCCL for the collectives.  The orbit statistic (ORCA subprocess) and NSPDK stay with the reference's CPU tools.
"""
import torch

from . import _lib as L
from .engine import stream_ptr

CLUSTER_BINS, SPECTRAL_BINS = 100, 200       # evaluation/stats.py:79, :133
SPECTRAL_RANGE = (-1e-5, 2.0)                # :133
_ISO = 5.0                                   # diagonal of removed (isolated) nodes: eigenvalue outside the range
# Eigenvalues of a normalised Laplacian lie in [0, 2], and 2 itself is common (every bipartite component has it).  The
# reference histograms LAPACK's values with range=(-1e-5, 2): a 2 that LAPACK returns as 2 + 4e-16 is silently dropped,
# one returned as 2 is counted -- a rounding accident of the host solver.  Here values within SNAP of the range are
# counted in the end bin, whatever the device solver's last bits are.  `snap=0.0` (argument of graph_histograms /
# eval_adjs) gives numpy's exact rule instead: a value outside [lo, hi] by any amount is dropped, as
# np.histogram(range=(-1e-5, 2)) does (evaluation/stats.py:133).  The two differ only for eigenvalues that the solver
# returns within 1e-9 ABOVE 2 (or below -1e-5): with the default, a bipartite component's eigenvalue 2 is always
# counted; with numpy's rule it is counted or not depending on the solver's last bit.
SNAP = 1e-9


def _adj(a, dev=None):
    a = torch.as_tensor(a)
    if dev is not None:
        a = a.to(dev)
    if a.device.type != "cuda":
        raise RuntimeError("swingnn_amd.graph_stats runs on an MI355X; there is no CPU fallback")
    a = a.to(torch.float32)
    if a.dim() != 3 or a.shape[1] != a.shape[2]:
        raise AssertionError("adjacency tensor must be [B, N, N]")
    return a.contiguous()


def graph_histograms(adjs, snap=SNAP):
    """[B,N,N] -> (degree histogram [B,N], clustering histogram [B,100], spectral pmf [B,200]), fp64 on the device.
    snap: tolerance of the spectral histogram's range ends (see SNAP; 0.0 = numpy's exact binning)."""
    a = _adj(adjs)
    B, N = a.shape[0], a.shape[1]
    f64 = dict(dtype=torch.float64, device=a.device)
    deg, clu = torch.empty((B, N), **f64), torch.empty((B, CLUSTER_BINS), **f64)
    lap = torch.empty((B, N, N), **f64)
    ws = torch.empty(max(int(L.lib.swg_graph_stats_workspace_bytes(B, N)), 256), dtype=torch.uint8, device=a.device)
    L.check(L.lib.swg_graph_hists(B, N, CLUSTER_BINS, a.data_ptr(), deg.data_ptr(), clu.data_ptr(), lap.data_ptr(),
                                  _ISO, ws.data_ptr(), ws.numel(), stream_ptr()), "swg_graph_hists")
    eig = torch.linalg.eigvalsh(lap).contiguous()       # [B, N] ascending; removed nodes give exactly _ISO
    pmf = torch.empty((B, SPECTRAL_BINS), **f64)
    L.check(L.lib.swg_hist_pmf(B, N, SPECTRAL_BINS, eig.data_ptr(), SPECTRAL_RANGE[0], SPECTRAL_RANGE[1], float(snap),
                               pmf.data_ptr(), stream_ptr()), "swg_hist_pmf")
    return deg, clu, pmf


def mmd_gaussian_tv(x, y, sigma=1.0):
    """compute_mmd(x, y, kernel=gaussian_tv, is_hist=True, sigma) for histogram matrices [n1,L1], [n2,L2] (zero padded
    to a common length as mmd.py:19-31 does) -> python float."""
    x, y = x.to(torch.float64), y.to(torch.float64)
    Lc = max(x.shape[1], y.shape[1])
    pad = lambda t: torch.nn.functional.pad(t, (0, Lc - t.shape[1])).contiguous()
    x, y = pad(x), pad(y)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    scratch = torch.empty(3 * (x.shape[0] + y.shape[0]), dtype=torch.float64, device=x.device)
    L.check(L.lib.swg_mmd_gaussian_tv(x.shape[0], y.shape[0], Lc, x.data_ptr(), y.data_ptr(), float(sigma),
                                      out.data_ptr(), scratch.data_ptr(), stream_ptr()), "swg_mmd_gaussian_tv")
    return float(out.item())


def eval_adjs(ref_adjs, pred_adjs, dev="cuda", snap=SNAP):
    """eval_torch_batch(ref, pred, methods=['degree', 'cluster', 'spectral']) of evaluation/stats.py:362-393 on
    adjacency tensors (lists of equally sized matrices or [B,N,N] tensors; the two sets may differ in N).
    snap: see SNAP (0.0 = numpy's exact range rule for the spectral histogram)."""
    stack = lambda s: s if torch.is_tensor(s) else torch.stack([torch.as_tensor(g, dtype=torch.float32) for g in s])
    r = graph_histograms(_adj(stack(ref_adjs), dev), snap)
    p = graph_histograms(_adj(stack(pred_adjs), dev), snap)
    out = {"degree": mmd_gaussian_tv(r[0], p[0], 1.0),      # stats.py:63
           "cluster": mmd_gaussian_tv(r[1], p[1], 0.1),     # :119
           "spectral": mmd_gaussian_tv(r[2], p[2], 1.0)}    # :170
    out["average"] = sum(out.values()) / 3.0                # :377
    return out
