"""GPU parity of the training step: loss and every parameter gradient of
``RainbowLoss(Precond(SwinGNN)(noisy, flags, sigmas), clean).mean().backward()`` from the HIP backward kernels
against autograd through the CPU oracle on the same weights and inputs (the oracle's forward is pinned to the
reference by tests/test_oracle_golden.py, its train-step loss / gradient fingerprints by the 'train/*' goldens).

Tolerance (bf16 MFMA operands in every Linear-layer product of the forward recompute and of the gradients, fp32
accumulation, fp32 LayerNorm / softmax / residual stream).  Per parameter tensor the error is
|g - g_ref|_2 / max(|g_ref|_2, 1e-3 |all gradients|_2):
  * 'init' regime (weights at the reference's initialisation scale, the regime training runs in): <= 3e-2
    (measured <= 1.6e-2), global gradient norm within 1 %;
  * 'stress' regime (unit-gain weights, O(1..10) activations; the forward itself is only held to 4e-2 of the output
    scale there and the error is carried back through up to 28 blocks): <= 0.3 (measured <= 0.21), cosine >= 0.97
    for every tensor above the floor, global norm within 3 %.
Loss within 1e-3 relative.
"""
import numpy as np
import pytest
import torch

from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NET_CASES, TRAIN_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

@pytest.mark.parametrize("name,regime,si,coin", TRAIN_CASES)
def test_train_step_gradients(golden_meta, name, regime, si, coin, monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import swingnn_amd as S
    from tests.test_gpu_parity import build_precond
    c = NET_CASES[name]
    cfg = c["cfg"]
    sd = H.net_state(golden_meta[f"keys/{name}"], c["seed"], regime)
    flags, x, sig, w, clean = H.train_case_inputs(c, si)
    ref_loss, ref = H.oracle_grads(sd, cfg, x, flags, sig, w, clean, coin)

    pre = build_precond(S, name, golden_meta, regime).train()
    loss_fn = S.RainbowLoss(regression_loss_weight=1.0, flag_reweight=False, objective="edm")
    monkeypatch.setattr(np.random, "rand", lambda: 0.25 if coin else 0.75)
    with torch.enable_grad():
        D = pre(x=x.to(DEV), node_flags=flags.to(DEV), sigmas=sig.to(DEV))
        loss = loss_fn(net_pred=D, net_target=clean.to(DEV), net_cond=sig.to(DEV), adjs_perturbed=x.to(DEV),
                       adjs_gt=clean.to(DEV), node_flags=flags.to(DEV), loss_weight=w.to(DEV), reduction='none')
        loss.mean().backward()
    pre.assert_symmetric()
    got_loss = loss.detach().cpu()
    assert torch.allclose(got_loss, ref_loss, rtol=1e-3, atol=1e-6), (got_loss, ref_loss)

    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
    tol, norm_tol = (3e-2, 1e-2) if regime == "init" else (0.3, 3e-2)
    got_total, bad = 0.0, []
    for k, p in pre.named_parameters():
        assert p.grad is not None, k
        g = p.grad.detach().float().cpu()
        assert torch.isfinite(g).all(), k
        got_total += float((g.double() ** 2).sum())
        r = ref[k]
        nr = float(r.norm())
        rel = float((g - r).norm()) / max(nr, 1e-3 * total)
        cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-30))
        if rel > tol or (nr > 1e-3 * total and cos < 0.97):
            bad.append((k, rel, cos, nr))
    assert not bad, bad
    assert abs(got_total ** 0.5 - total) <= norm_tol * total, (got_total ** 0.5, total)
