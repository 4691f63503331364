"""GPU parity of the molecule path (SURVEY 8 row a12 + the NodeAdj halves of a13 / a15): NodeAdjSwinGNN,
NodeAdjPrecond and NodeAdjEDMSampler through swg_nodeadj_forward / swg_heun_nodes_*, against the golden vectors
produced by the reference modules and against the CPU oracle.

Tolerances as in test_gpu_parity.py: 'init' regime -> max-abs error on the denoised adjacency / node features
<= 1e-2 (north-star bar); 'stress' regime -> error relative to the output scale (max 4e-2, mean 5e-3)."""
import numpy as np
import pytest
import torch

from oracle import filler
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NODEADJ_CASES, NODEADJ_SAMPLER_CASES
from tests.test_gpu_parity import DEV, load, rel_close

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import swingnn_amd
    return swingnn_amd


def build(S, name, meta, regime, seed):
    cfg = NODEADJ_CASES[name]["cfg"]
    net = S.NodeAdjSwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                           embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                           window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                           drop_path_rate=0.0, self_condition=cfg["self_condition"], symmetric_noise=True,
                           out_chans_adj=cfg["out_chans_adj"], out_chans_node=cfg["out_chans_node"])
    pre = S.NodeAdjPrecond("edm", net, cfg["self_condition"], symmetric_noise=True)
    return load(pre, H.net_state(meta[f"keys/{name}"], seed, regime))


def check(out, ref, regime):
    if regime == "init":
        out = out.detach().float().cpu()
        ref = torch.as_tensor(ref)
        assert out.shape == ref.shape, (out.shape, ref.shape)
        err = float((out - ref).abs().max())
        assert err <= 1e-2, f"max abs err {err:.4g}"
    else:
        rel_close(out, ref)


@pytest.mark.parametrize("name", list(NODEADJ_CASES))
def test_nodeadj_network_and_precond(S, golden_nodeadj, name):
    g, meta = golden_nodeadj
    c = NODEADJ_CASES[name]
    cfg = c["cfg"]
    B, N = c["B"], cfg["img_size"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    d = lambda t: t.to(DEV)
    for regime in c["regimes"]:
        pre = build(S, name, meta, regime, c["seed"])
        assert sorted(pre.state_dict().keys()) == sorted(meta[f"keys/{name}"].keys())
        for si, sigma in enumerate(c["sigmas"]):
            adj, node, sca, scn, sig = H.nodeadj_inputs(cfg, B, flags, c["seed"] + si, sigma)
            key = f"na/{name}/{regime}/s{si}/"
            c_noise = sig.log() / 4
            a, n = pre.model(d(adj), d(node), d(flags), d(c_noise), None, None)
            check(a, g[key + "Fa_nosc"], regime); check(n, g[key + "Fn_nosc"], regime)
            a, n = pre.model(d(adj), d(node), d(flags), d(c_noise), d(sca), d(scn))
            check(a, g[key + "Fa_sc"], regime); check(n, g[key + "Fn_sc"], regime)
            for coin, u in (("coin1", 0.25), ("coin0", 0.75)):
                saved = np.random.rand
                np.random.rand = lambda u=u: u
                try:
                    a, n = pre(d(adj), d(node), d(flags), d(sig), d(sca), d(scn))
                finally:
                    np.random.rand = saved
                check(a, g[key + f"Da_{coin}"], regime); check(n, g[key + f"Dn_{coin}"], regime)
                assert float((a - a.transpose(-1, -2)).abs().max()) == 0.0  # exactly symmetric, like the reference
            pre.assert_symmetric()


@pytest.mark.parametrize("name", list(NODEADJ_SAMPLER_CASES))
def test_nodeadj_sampler(S, golden_nodeadj, name):
    g, meta = golden_nodeadj
    c = NODEADJ_SAMPLER_CASES[name]
    cfg = NODEADJ_CASES[c["net"]]["cfg"]
    pre = build(S, c["net"], meta, c["regime"], c["seed"])
    smp = S.NodeAdjEDMSampler(num_steps=c["steps"], clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                              clip_samples_scope="x_0", self_condition=cfg["self_condition"], dev=DEV,
                              symmetric_noise=True)
    B, N = c["B"], cfg["img_size"]
    Ca, Cn = cfg["out_chans_adj"], cfg["out_chans_node"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    gta = gtn = None
    if c["sanity"]:
        gta, gtn, _, _, _ = H.nodeadj_inputs(cfg, B, flags, c["seed"], 0.0)
        gta, gtn = gta.sign().to(DEV), gtn.sign().to(DEV)
    # the init draw comes from the CPU generator in the reference's order (adjacency, then nodes)
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    ia, inn = smp.gen_init_sample(flags.to(DEV), folded_norm=False, num_node_chan=Cn, num_edge_chan=Ca)
    assert H.max_abs(ia.cpu(), g[f"nasampler/{name}/init_adjs"]) == 0.0
    assert H.max_abs(inn.cpu(), g[f"nasampler/{name}/init_nodes"]) == 0.0
    na, nn_ = torch.as_tensor(g[f"nasampler/{name}/noise_adjs"]), torch.as_tensor(g[f"nasampler/{name}/noise_nodes"])
    it = iter([t for pair in zip(na, nn_) for t in pair])
    smp._randn_like = lambda x: next(it).to(x.device).reshape(x.shape).contiguous()
    fa, fn = smp.sample(pre, flags.to(DEV), init_adjs=ia, init_nodes=inn, sanity_check_gt_adjs=gta,
                        sanity_check_gt_nodes=gtn, num_node_chan=Cn, num_edge_chan=Ca)
    ra, rn = torch.as_tensor(g[f"nasampler/{name}/final_adjs"]), torch.as_tensor(g[f"nasampler/{name}/final_nodes"])
    assert fa.device.type == "cpu" and fa.shape == ra.shape and fn.shape == rn.shape
    tol = 1e-5 if c["sanity"] else 1e-2
    ea, en = H.max_abs(fa, ra), H.max_abs(fn, rn)
    assert ea <= tol and en <= tol, f"final sample max abs err adj {ea:.4g} nodes {en:.4g}"
    if c["sanity"]:
        assert H.max_abs(fa, O.mask_adjs(gta.cpu(), flags)) < 1e-5 and H.max_abs(fn, O.mask_nodes(gtn.cpu(), flags)) < 1e-5
    print(f"{name}: final max err adj {ea:.3g} nodes {en:.3g}")
