"""Golden vectors of the TRAIN STEP from the REFERENCE's own modules on CPU:
loss [B], global gradient norm and per-parameter gradient fingerprints (sum, abs-sum, and the gradient itself for
a few small tensors) of ``RainbowLoss(Precond(SwinGNN)(x, flags, sigmas), clean).mean().backward()``
(runner/trainer/trainer.py:108-135).

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py

Same conventions as make_golden.py: only data is written (golden_train.npz / golden_train_meta.json); weights and
inputs are regenerated on both sides by oracle/filler.py; the only shim is the 3-name in-memory timm.models.layers.
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

from model.swin_gnn import swin_gnn as R          # noqa: E402  (reference)
from model.precond.precond import Precond as RPrecond  # noqa: E402
from loss.rainbow_loss import RainbowLoss as RLoss  # noqa: E402

from oracle import filler  # noqa: E402
from tests.golden.cases import NET_CASES, TRAIN_CASES  # noqa: E402

OUT, META = {}, {}
SMALL = ("norm.weight", "read_out.2.weight", "map_layer0.bias", "relative_position_bias_table")

for name, regime, si, coin in TRAIN_CASES:
    c = NET_CASES[name]
    cfg = c["cfg"]
    net = R.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                    window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=cfg["self_condition"])
    pre = RPrecond("edm", net, cfg["self_condition"]).train()
    pre.load_state_dict(filler.fill_state_dict(pre.state_dict(), c["seed"], regime), strict=True)
    B, N = c["B"], cfg["img_size"]
    sigma = c["sigmas"][si]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    x = filler.sym_adj("x", B, N, flags, c["seed"] + si, scale=float(np.sqrt(sigma ** 2 + 0.25)))
    sig = torch.full((B,), float(sigma)) * (1.0 + 0.1 * filler.uniform("sig", (B,), c["seed"])).abs()
    clean = torch.sign(x) * flags[:, :, None] * flags[:, None, :]
    w = (sig ** 2 + 0.25) / (sig * 0.5) ** 2
    saved = np.random.rand
    np.random.rand = lambda: 0.25 if coin else 0.75
    try:
        D = pre(x=x.clone(), node_flags=flags, sigmas=sig)
    finally:
        np.random.rand = saved
    loss = RLoss(regression_loss_weight=1.0, flag_reweight=False, objective="edm")(
        net_pred=D, net_target=clean, net_cond=sig, adjs_perturbed=x, adjs_gt=clean, node_flags=flags,
        loss_weight=w, reduction="none")
    loss.mean().backward()
    key = f"train/{name}/{regime}/s{si}/coin{int(coin)}"
    names = [k for k, p in pre.named_parameters()]
    grads = [p.grad.double() for _, p in pre.named_parameters()]
    OUT[key + "/loss"] = loss.detach().numpy()
    OUT[key + "/grad_norm"] = np.array(float(torch.sqrt(sum((g ** 2).sum() for g in grads))))
    OUT[key + "/grad_sums"] = np.array([float(g.sum()) for g in grads])
    OUT[key + "/grad_abs_sums"] = np.array([float(g.abs().sum()) for g in grads])
    for k, g in zip(names, grads):
        if k.endswith(SMALL) and g.numel() <= 4096:
            OUT[f"{key}/grad/{k}"] = g.float().numpy()
    META["param_names/" + name] = names

np.savez_compressed(os.path.join(HERE, "golden_train.npz"), **OUT)
with open(os.path.join(HERE, "golden_train_meta.json"), "w") as f:
    json.dump(META, f, indent=0, sort_keys=True)
print({k: v.shape for k, v in OUT.items() if k.endswith(("loss", "grad_norm"))})
