"""Pin the CPU oracle of the molecule (node + adjacency) path against golden vectors produced by the reference's
NodeAdjSwinGNN / NodeAdjPrecond / NodeAdjEDMSampler.  fp32 vs fp32 on the same CPU: 2e-5 relative to the scale."""
import pytest
import torch

from oracle import filler
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NODEADJ_CASES, NODEADJ_SAMPLER_CASES

torch.set_grad_enabled(False)


def close(a, ref, rtol=2e-5):
    ref = torch.as_tensor(ref)
    assert tuple(a.shape) == tuple(ref.shape), f"shape {tuple(a.shape)} vs golden {tuple(ref.shape)}"
    scale = max(1.0, float(ref.abs().max()))
    err = H.max_abs(a, ref)
    assert err <= rtol * scale, f"max abs err {err:.3e} (scale {scale:.3g})"


@pytest.mark.parametrize("name", list(NODEADJ_CASES))
def test_nodeadj_network_and_precond(golden_nodeadj, name):
    g, meta = golden_nodeadj
    c = NODEADJ_CASES[name]
    cfg = c["cfg"]
    B, N = c["B"], cfg["img_size"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    for regime in c["regimes"]:
        sd = H.net_state(meta[f"keys/{name}"], c["seed"], regime)
        net = O._strip(sd)
        for si, sigma in enumerate(c["sigmas"]):
            adj, node, sca, scn, sig = H.nodeadj_inputs(cfg, B, flags, c["seed"] + si, sigma)
            key = f"na/{name}/{regime}/s{si}/"
            a, n = O.nodeadj_forward(net, cfg, adj, node, flags, sig.log() / 4, None, None)
            close(a, g[key + "Fa_nosc"]); close(n, g[key + "Fn_nosc"])
            a, n = O.nodeadj_forward(net, cfg, adj, node, flags, sig.log() / 4, sca, scn)
            close(a, g[key + "Fa_sc"]); close(n, g[key + "Fn_sc"])
            for coin in (True, False):
                a, n = O.nodeadj_precond_forward(sd, cfg, adj, node, flags, sig, coin, sca, scn)
                close(a, g[key + f"Da_coin{int(coin)}"]); close(n, g[key + f"Dn_coin{int(coin)}"])


@pytest.mark.parametrize("name", list(NODEADJ_SAMPLER_CASES))
def test_nodeadj_sampler(golden_nodeadj, name):
    import numpy as np
    g, meta = golden_nodeadj
    c = NODEADJ_SAMPLER_CASES[name]
    cfg = NODEADJ_CASES[c["net"]]["cfg"]
    sd = H.net_state(meta[f"keys/{c['net']}"], c["seed"], c["regime"])
    B, N = c["B"], cfg["img_size"]
    Ca, Cn = cfg["out_chans_adj"], cfg["out_chans_node"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    # the init draw of gen_init_sample comes from the CPU generator: reproduce it from the seed
    torch.manual_seed(c["seed"])
    ia, inn = O.nodeadj_gen_init_sample(flags, Cn, Ca)
    close(ia, g[f"nasampler/{name}/init_adjs"], 0); close(inn, g[f"nasampler/{name}/init_nodes"], 0)
    na, nn_ = torch.as_tensor(g[f"nasampler/{name}/noise_adjs"]), torch.as_tensor(g[f"nasampler/{name}/noise_nodes"])
    it = iter([t for pair in zip(na, nn_) for t in pair])
    gt = None
    if c["sanity"]:
        gta, gtn, _, _, _ = H.nodeadj_inputs(cfg, B, flags, c["seed"], 0.0)
        gt = (gta.sign(), gtn.sign())
    np.random.seed(c["seed"])

    def den(a, n, f, s, sa, sn):
        return O.nodeadj_precond_forward(sd, cfg, a, n, f, s, np.random.rand() < 0.5, sa, sn)

    fa, fn = O.nodeadj_edm_sample(den, flags, ia, inn, num_steps=c["steps"], self_condition=cfg["self_condition"],
                                  noise_fn=lambda x: next(it), sanity_gt=gt)
    close(fa, g[f"nasampler/{name}/final_adjs"], 1e-4); close(fn, g[f"nasampler/{name}/final_nodes"], 1e-4)
    if c["sanity"]:  # known-answer mode: a perfect denoiser reproduces the ground truth at t = 0
        close(fa, O.mask_adjs(gt[0], flags), 1e-5); close(fn, O.mask_nodes(gt[1], flags), 1e-5)


def test_nodeadj_objective(golden_nodeadj_train):
    """NodeAdjEDMObjectiveGenerator restated with the oracle's pieces, same CPU-generator draws in the same order."""
    g = golden_nodeadj_train
    for name in ("qm9_ddpm60", "qm9_onehot96"):
        cfg = NODEADJ_CASES[name]["cfg"]
        B, N = 5, cfg["img_size"]
        flags = filler.node_flags("flags", B, N, 3, 9, 7)
        ca, cx, _, _, _ = H.nodeadj_inputs(cfg, B, flags, 7, 0.0)
        ca, cx = ca.sign(), cx.sign()
        torch.manual_seed(7)
        sig, w = O.edm_training_sigmas_weights(torch.randn(B))
        raw, raw_x = torch.randn_like(ca), torch.randn_like(cx)
        noisy = O.mask_adjs(ca + O.sym_noise_from_raw(raw) * sig.reshape([-1] + [1] * (ca.dim() - 1)), flags)
        noisy_x = cx + O.mask_nodes(raw_x * sig.reshape([-1] + [1] * (cx.dim() - 1)), flags)
        close(sig, g[f"naobj/{name}/sigmas"], 1e-6); close(w, g[f"naobj/{name}/weights"], 1e-6)
        close(noisy, g[f"naobj/{name}/noisy_adjs"], 1e-6); close(noisy_x, g[f"naobj/{name}/noisy_nodes"], 1e-6)


@pytest.mark.parametrize("name,regime,si,coin", __import__("tests.golden.cases", fromlist=["x"]).NODEADJ_TRAIN_CASES)
def test_nodeadj_train_step_fingerprints(golden_nodeadj, golden_nodeadj_train, name, regime, si, coin):
    """Autograd through the oracle reproduces the reference's losses and gradient fingerprints."""
    import numpy as np
    _, meta = golden_nodeadj
    g = golden_nodeadj_train
    c = NODEADJ_CASES[name]
    sd = H.net_state(meta[f"keys/{name}"], c["seed"], regime)
    flags, adj, node, sig, w, clean_a, clean_x = H.nodeadj_train_inputs(c, si)
    with torch.enable_grad():
        la, ln, grads = H.oracle_grads_nodeadj(sd, c["cfg"], adj, node, flags, sig, w, clean_a, clean_x, coin)
    key = f"natrain/{name}/{regime}/s{si}/coin{int(coin)}"
    assert np.allclose(la.numpy(), g[key + "/loss_adj"], rtol=1e-4) and np.allclose(ln.numpy(), g[key + "/loss_node"], rtol=1e-4)
    names = [k for k, v in meta[f"keys/{name}"].items()]
    order = [k for k in sd if k in grads]
    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
    assert abs(total - float(g[key + "/grad_norm"])) <= 2e-4 * float(g[key + "/grad_norm"])
    assert len(g[key + "/grad_abs_sums"]) == len(grads)
    mine = np.sort(np.array([float(v.double().abs().sum()) for v in grads.values()]))
    assert np.allclose(mine, np.sort(g[key + "/grad_abs_sums"]), rtol=2e-3, atol=1e-6 * total)
