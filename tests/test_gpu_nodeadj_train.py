"""GPU parity of the molecule train step (runner/trainer/trainer_node_adj.py:95-126): objective, the two losses and
every parameter gradient of ``(loss_adj.mean() + loss_node.mean()).backward()`` from the HIP backward against autograd
through the CPU oracle (pinned to the reference's losses / gradient fingerprints by tests/test_oracle_nodeadj.py).
Tolerances as in test_gpu_train.py: per tensor |g - g_ref|_2 / max(|g_ref|_2, 1e-3 |all|_2) <= 3e-2 ('init') / 0.3
('stress', cosine >= 0.97), global norm within 1 % / 3 %, losses within 1e-3 ('init') / 5e-3 ('stress': the forward
itself is only held to 4e-2 of the output scale there) relative."""
import numpy as np
import pytest
import torch

from oracle import filler
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NODEADJ_CASES, NODEADJ_TRAIN_CASES
from tests.test_gpu_nodeadj import build

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name,regime,si,coin", NODEADJ_TRAIN_CASES)
def test_nodeadj_train_step_gradients(golden_nodeadj, name, regime, si, coin, monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import swingnn_amd as S
    _, meta = golden_nodeadj
    c = NODEADJ_CASES[name]
    sd = H.net_state(meta[f"keys/{name}"], c["seed"], regime)
    flags, adj, node, sig, w, clean_a, clean_x = H.nodeadj_train_inputs(c, si)
    ref_la, ref_ln, ref = H.oracle_grads_nodeadj(sd, c["cfg"], adj, node, flags, sig, w, clean_a, clean_x, coin)

    pre = build(S, name, meta, regime, c["seed"]).train()
    loss_fn = S.NodeAdjRainbowLoss(edge_loss_weight=1.0, node_loss_weight=1.0, objective="edm")
    monkeypatch.setattr(np.random, "rand", lambda: 0.25 if coin else 0.75)
    d = lambda t: t.to(DEV)
    with torch.enable_grad():
        Da, Dn = pre(adjs=d(adj), nodes=d(node), node_flags=d(flags), sigmas=d(sig))
        la, ln = loss_fn(net_pred_a=Da, net_pred_x=Dn, net_target_a=d(clean_a), net_target_x=d(clean_x), net_cond=d(sig),
                         adjs_perturbed=d(adj), adjs_gt=d(clean_a), x_perturbed=d(node), x_gt=d(clean_x),
                         node_flags=d(flags), loss_weight=d(w), reduction='none')
        (la.mean() + ln.mean()).backward()
    pre.assert_symmetric()
    ltol = 1e-3 if regime == "init" else 5e-3
    assert torch.allclose(la.detach().cpu(), ref_la, rtol=ltol, atol=1e-6), (la, ref_la)
    assert torch.allclose(ln.detach().cpu(), ref_ln, rtol=ltol, atol=1e-6), (ln, ref_ln)
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


def test_nodeadj_objective_and_node_loss(golden_nodeadj_train):
    """NodeAdjEDMObjectiveGenerator with the reference's CPU-generator draws, against the reference goldens."""
    import swingnn_amd as S
    g = golden_nodeadj_train
    for name in ("qm9_ddpm60", "qm9_onehot96"):
        cfg = NODEADJ_CASES[name]["cfg"]
        B, N = 5, cfg["img_size"]
        flags = filler.node_flags("flags", B, N, 3, 9, 7)
        ca, cx, _, _, _ = H.nodeadj_inputs(cfg, B, flags, 7, 0.0)
        ca, cx = ca.sign(), cx.sign()
        obj = S.NodeAdjEDMObjectiveGenerator("edm", "edm", other_params=None, dev=DEV, symmetric_noise=True)
        torch.manual_seed(7)
        obj._draw = lambda clean, x: tuple(t.to(DEV) for t in (torch.randn(clean.shape[0]), torch.randn(clean.shape),
                                                               torch.randn(x.shape)))
        na, nx, cond, ta, tx, (c_skip, c_out, c_in, c_noise, sigmas, weights) = obj.get_input_output(
            ca.to(DEV), cx.to(DEV), flags.to(DEV))
        for got, key in ((na, "noisy_adjs"), (nx, "noisy_nodes"), (sigmas, "sigmas"), (weights, "weights")):
            ref = torch.as_tensor(g[f"naobj/{name}/{key}"])
            assert got.shape == ref.shape and torch.allclose(got.cpu(), ref, rtol=2e-6, atol=2e-6), key
        cs, co, ci, cn = O.edm_coefficients(torch.as_tensor(g[f"naobj/{name}/sigmas"]))
        assert torch.allclose(c_skip.cpu(), cs, rtol=1e-5) and torch.allclose(c_noise.cpu(), cn, rtol=1e-5, atol=1e-6)


def test_nodeadj_train_steps_reduce_loss(golden_nodeadj, monkeypatch):
    """Full molecule train steps (objective -> NodeAdjPrecond forward -> two losses -> backward -> clip + Adam + EMA,
    runner/trainer/trainer_node_adj.py:79-131) on one batch: the loss on that batch goes down."""
    import swingnn_amd as S
    _, meta = golden_nodeadj
    c = NODEADJ_CASES["qm9_onehot96"]
    pre = build(S, "qm9_onehot96", meta, "init", c["seed"]).train()
    flags, adj, node, sig, w, clean_a, clean_x = (t.to(DEV) for t in H.nodeadj_train_inputs(c, 0))
    opt = S.FusedAdamEMA(pre.parameters(), lr=1e-3, max_norm=10.0, ema_betas=(0.9, 0.99))
    loss_fn = S.NodeAdjRainbowLoss(edge_loss_weight=1.0, node_loss_weight=1.0, objective="edm")
    monkeypatch.setattr(np.random, "rand", lambda: 0.75)
    losses = []
    with torch.enable_grad():
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            Da, Dn = pre(adjs=adj, nodes=node, node_flags=flags, sigmas=sig)
            la, ln = loss_fn(Da, Dn, clean_a, clean_x, sig, adj, clean_a, node, clean_x, flags, loss_weight=w,
                             reduction='none')
            loss = la.mean() + ln.mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
