"""The no-alloc entry point (swg_edm_loss_noalloc: partial sums in a file-scope device array of the library, no
workspace argument) against the existing entry point swg_edm_loss through the C-ABI.  Bit-exact: same kernels, same
summation order.  Two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import pytest
import torch

from oracle import filler
from oracle import swingnn_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, N, seed):
    flags = filler.node_flags("flags", B, N, max(2, N // 4), N - 1, seed)
    clean = filler.sym_adj("clean", B, N, flags, seed).sign()
    pred = filler.sym_adj("pred", B, N, flags, seed + 1)
    w = filler.uniform("w", (B,), seed).abs() + 0.5
    return flags, clean, pred, w


def test_edm_loss_noalloc_matches_edm_loss():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from swingnn_amd import _lib as L, engine as E
    d = lambda t: t.to(DEV).contiguous()
    sp = E.stream_ptr()
    for B, N, seed in ((37, 48, 3), (5, 24, 4)):   # second call smaller than the first
        flags, clean, pred, w = _case(B, N, seed)
        fl, cl, pr, wd = d(flags), d(clean), d(pred), d(w)
        for mean in (0, 1):
            want = torch.empty(B, device=DEV)
            scratch = torch.empty(2 * B, device=DEV)
            L.check(L.lib.swg_edm_loss(B, N, pr.data_ptr(), cl.data_ptr(), fl.data_ptr(), wd.data_ptr(), 1.5, mean,
                                       want.data_ptr(), scratch.data_ptr(), sp), "swg_edm_loss")
            got = torch.full((B,), float("nan"), device=DEV)
            L.check(L.lib.swg_edm_loss_noalloc(B, N, pr.data_ptr(), cl.data_ptr(), fl.data_ptr(), wd.data_ptr(), 1.5,
                                               mean, got.data_ptr(), sp), "swg_edm_loss_noalloc")
            assert torch.equal(got, want), (B, N, mean, float((got - want).abs().max()))
            if not mean:
                assert torch.allclose(got.cpu(), 1.5 * O.edm_loss(pred, clean, flags, w), rtol=1e-5)
    # NULL weights, and the size limit
    got = torch.empty(5, device=DEV)
    L.check(L.lib.swg_edm_loss_noalloc(5, 24, pr.data_ptr(), cl.data_ptr(), fl.data_ptr(), None, 1.0, 0,
                                       got.data_ptr(), sp), "swg_edm_loss_noalloc")
    assert torch.allclose(got.cpu(), O.edm_loss(pred, clean, flags, torch.ones(5)), rtol=1e-5)
    assert L.lib.swg_edm_loss_noalloc(65537, 24, pr.data_ptr(), cl.data_ptr(), fl.data_ptr(), None, 1.0, 0,
                                      got.data_ptr(), sp) == -3
