"""Train-step goldens of the molecule path from the REFERENCE's own modules on CPU: NodeAdjEDMObjectiveGenerator
(runner/objectives/edm.py:215-276) under a fixed torch seed, and loss [B] x 2 + gradient fingerprints of
``(loss_adj.mean() + loss_node.mean()).backward()`` through NodeAdjRainbowLoss(NodeAdjPrecond(NodeAdjSwinGNN))
(runner/trainer/trainer_node_adj.py:95-126).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nodeadj_train.py     (build container only)
"""
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

from model.swin_gnn import swin_gnn as R                                          # noqa: E402  (reference)
from model.precond.precond import NodeAdjPrecond as RPrecond                      # noqa: E402
from runner.objectives.edm import NodeAdjEDMObjectiveGenerator as RObj            # noqa: E402
from loss.rainbow_loss import NodeAdjRainbowLoss as RLoss                         # noqa: E402

from oracle import filler                                                         # noqa: E402
from tests.golden.cases import NODEADJ_CASES, NODEADJ_TRAIN_CASES                # noqa: E402
from tests.helpers import nodeadj_inputs, nodeadj_train_inputs                    # noqa: E402

OUT = {}

# ---- objective (draws: sigmas, adjacency noise, node noise from the CPU generator under seed 7) ----
for name in ("qm9_ddpm60", "qm9_onehot96"):
    cfg = NODEADJ_CASES[name]["cfg"]
    B, N = 5, cfg["img_size"]
    flags = filler.node_flags("flags", B, N, 3, 9, 7)
    clean_a, clean_x, _, _, _ = nodeadj_inputs(cfg, B, flags, 7, 0.0)
    clean_a, clean_x = clean_a.sign(), clean_x.sign()
    obj = RObj("edm", "edm", other_params=None, dev="cpu", symmetric_noise=True)
    torch.manual_seed(7)
    with torch.no_grad():
        na, nx, cond, ta, tx, (c_skip, c_out, c_in, c_noise, sigmas, weights) = obj.get_input_output(
            clean_a.clone(), clean_x.clone(), flags)
    OUT[f"naobj/{name}/noisy_adjs"], OUT[f"naobj/{name}/noisy_nodes"] = na.numpy(), nx.numpy()
    OUT[f"naobj/{name}/sigmas"], OUT[f"naobj/{name}/weights"] = sigmas.numpy(), weights.numpy()

# ---- train step ----
for name, regime, si, coin in NODEADJ_TRAIN_CASES:
    c = NODEADJ_CASES[name]
    cfg = c["cfg"]
    net = R.NodeAdjSwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                           embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                           window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                           drop_path_rate=0.0, self_condition=cfg["self_condition"], symmetric_noise=True,
                           out_chans_adj=cfg["out_chans_adj"], out_chans_node=cfg["out_chans_node"])
    pre = RPrecond("edm", net, cfg["self_condition"], symmetric_noise=True).train()
    pre.load_state_dict(filler.fill_state_dict(pre.state_dict(), c["seed"], regime), strict=True)
    flags, adj, node, sig, w, clean_a, clean_x = nodeadj_train_inputs(c, si)
    saved = np.random.rand
    np.random.rand = lambda: 0.25 if coin else 0.75
    try:
        Da, Dn = pre(adjs=adj.clone(), nodes=node.clone(), node_flags=flags, sigmas=sig)
    finally:
        np.random.rand = saved
    la, ln = RLoss(edge_loss_weight=1.0, node_loss_weight=1.0, objective="edm")(
        net_pred_a=Da, net_pred_x=Dn, net_target_a=clean_a, net_target_x=clean_x, net_cond=sig, adjs_perturbed=adj,
        adjs_gt=clean_a, x_perturbed=node, x_gt=clean_x, node_flags=flags, loss_weight=w, reduction="none")
    (la.mean() + ln.mean()).backward()
    key = f"natrain/{name}/{regime}/s{si}/coin{int(coin)}"
    grads = [p.grad.double() for _, p in pre.named_parameters()]
    OUT[key + "/loss_adj"], OUT[key + "/loss_node"] = la.detach().numpy(), ln.detach().numpy()
    OUT[key + "/grad_norm"] = np.array(float(torch.sqrt(sum((g ** 2).sum() for g in grads))))
    OUT[key + "/grad_sums"] = np.array([float(g.sum()) for g in grads])
    OUT[key + "/grad_abs_sums"] = np.array([float(g.abs().sum()) for g in grads])

np.savez_compressed(os.path.join(HERE, "golden_nodeadj_train.npz"), **OUT)
print({k: v.shape for k, v in OUT.items() if k.endswith(("loss_adj", "grad_norm", "sigmas"))})
