"""Golden vectors of the molecule (node + adjacency) path, from the REFERENCE's own modules on CPU:
NodeAdjSwinGNN (model/swin_gnn/swin_gnn.py:782-1021), NodeAdjPrecond (model/precond/precond.py:60-114) and
NodeAdjEDMSampler (runner/mcmc_sampler/edm.py:235-449).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nodeadj.py      (build container only)

Committed: expected outputs (golden_nodeadj.npz) and the key tables (golden_nodeadj_meta.json); weights and inputs
are regenerated on both sides by oracle/filler.py.  Same 3-name timm shim as make_golden.py.  The sampler cases
record the raw normals that ``torch.randn_like`` returned (CPU generator), in call order, so that the device path
can be driven with identical noise.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

timm = types.ModuleType("timm")
timm_models = types.ModuleType("timm.models")
timm_layers = types.ModuleType("timm.models.layers")


class DropPath(torch.nn.Identity):
    def __init__(self, drop_prob=0.0):
        super().__init__()


timm_layers.DropPath = DropPath
timm_layers.to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
timm.models = timm_models
timm_models.layers = timm_layers
sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})
sys.path.insert(0, "/root/reference")

from model.swin_gnn import swin_gnn as R                                  # noqa: E402  (reference)
from model.precond.precond import NodeAdjPrecond as RPrecond              # noqa: E402
from runner.mcmc_sampler.edm import NodeAdjEDMSampler as RSampler         # noqa: E402

from oracle import filler                                                 # noqa: E402
from tests.golden.cases import NODEADJ_CASES, NODEADJ_SAMPLER_CASES      # noqa: E402
from tests.helpers import nodeadj_inputs                                  # noqa: E402

torch.set_grad_enabled(False)
OUT, META = {}, {}


def build(cfg):
    net = R.NodeAdjSwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                           embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                           window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                           drop_path_rate=0.0, self_condition=cfg["self_condition"], symmetric_noise=True,
                           out_chans_adj=cfg["out_chans_adj"], out_chans_node=cfg["out_chans_node"]).eval()
    return net, RPrecond("edm", net, cfg["self_condition"], symmetric_noise=True).eval()


def fill(module, seed, regime):
    module.load_state_dict(filler.fill_state_dict(module.state_dict(), seed, regime), strict=True)
    return module


for name, c in NODEADJ_CASES.items():
    cfg = c["cfg"]
    net, pre = build(cfg)
    META[f"keys/{name}"] = {k: [list(v.shape), str(v.dtype)] for k, v in pre.state_dict().items()}
    META[f"nparams/{name}"] = sum(p.numel() for p in pre.parameters())
    B, N = c["B"], cfg["img_size"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    for regime in c["regimes"]:
        fill(pre, c["seed"], regime)
        for si, sigma in enumerate(c["sigmas"]):
            adj, node, sca, scn, sig = nodeadj_inputs(cfg, B, flags, c["seed"] + si, sigma)
            c_noise = sig.log() / 4
            a, n = net(adj.clone(), node.clone(), flags, c_noise, None, None)
            OUT[f"na/{name}/{regime}/s{si}/Fa_nosc"], OUT[f"na/{name}/{regime}/s{si}/Fn_nosc"] = a.numpy(), n.numpy()
            a, n = net(adj.clone(), node.clone(), flags, c_noise, sca.clone(), scn.clone())
            OUT[f"na/{name}/{regime}/s{si}/Fa_sc"], OUT[f"na/{name}/{regime}/s{si}/Fn_sc"] = a.numpy(), n.numpy()
            for coin, u in (("coin1", 0.25), ("coin0", 0.75)):
                saved = np.random.rand
                np.random.rand = lambda u=u: u
                try:  # NodeAdjPrecond uses the passed self-conditioning tensors unless the coin replaces them
                    a, n = pre(adj.clone(), node.clone(), flags, sig, sca.clone(), scn.clone())
                finally:
                    np.random.rand = saved
                OUT[f"na/{name}/{regime}/s{si}/Da_{coin}"], OUT[f"na/{name}/{regime}/s{si}/Dn_{coin}"] = a.numpy(), n.numpy()

for name, c in NODEADJ_SAMPLER_CASES.items():
    cfg = NODEADJ_CASES[c["net"]]["cfg"]
    net, pre = build(cfg)
    fill(pre, c["seed"], c["regime"])
    smp = RSampler(num_steps=c["steps"], clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                   clip_samples_scope="x_0", self_condition=cfg["self_condition"], dev="cpu", symmetric_noise=True)
    B, N = c["B"], cfg["img_size"]
    Ca, Cn = cfg["out_chans_adj"], cfg["out_chans_node"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    gta = gtn = None
    if c["sanity"]:
        gta, gtn, _, _, _ = nodeadj_inputs(cfg, B, flags, c["seed"], 0.0)
        gta, gtn = gta.sign(), gtn.sign()
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    init_a, init_n = smp.gen_init_sample(flags, folded_norm=False, num_node_chan=Cn, num_edge_chan=Ca)
    OUT[f"nasampler/{name}/init_adjs"], OUT[f"nasampler/{name}/init_nodes"] = init_a.numpy(), init_n.numpy()
    drawn = []
    real_randn_like = torch.randn_like

    def rec(x, *a, **k):
        r = real_randn_like(x, *a, **k)
        drawn.append(r.clone())
        return r

    torch.randn_like = rec
    try:
        fa, fn = smp.sample(pre, flags, init_adjs=init_a.clone(), init_nodes=init_n.clone(), sanity_check_gt_adjs=gta,
                            sanity_check_gt_nodes=gtn, num_node_chan=Cn, num_edge_chan=Ca)
    finally:
        torch.randn_like = real_randn_like
    assert len(drawn) == 2 * c["steps"]
    OUT[f"nasampler/{name}/noise_adjs"] = torch.stack(drawn[0::2]).numpy()
    OUT[f"nasampler/{name}/noise_nodes"] = torch.stack(drawn[1::2]).numpy()
    OUT[f"nasampler/{name}/final_adjs"], OUT[f"nasampler/{name}/final_nodes"] = fa.numpy(), fn.numpy()

# seeded initialisation fingerprint (module creation order + init calls)
cfg = NODEADJ_CASES["qm9_onehot96"]["cfg"]
torch.manual_seed(1234)
net, pre = build(cfg)
sd = pre.state_dict()
keys = [k for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith("attn_mask")]
OUT["nainit/qm9_onehot96/sums"] = np.array([float(sd[k].double().sum()) for k in keys])
OUT["nainit/qm9_onehot96/abs_sums"] = np.array([float(sd[k].double().abs().sum()) for k in keys])
META["init_keys/qm9_onehot96"] = keys

np.savez_compressed(os.path.join(HERE, "golden_nodeadj.npz"), **OUT)
with open(os.path.join(HERE, "golden_nodeadj_meta.json"), "w") as f:
    json.dump(META, f, indent=0, sort_keys=True)
print("wrote", len(OUT), "arrays;", sum(v.nbytes for v in OUT.values()) / 1e6, "MB raw")
