"""Generate golden vectors by running the REFERENCE's own modules on CPU.

Run in the build container only (needs /root/reference; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is committed is data: expected outputs (``*.npz``) and the key/shape tables
(``*.json``).  Weights and inputs are NOT stored; both sides regenerate them with
``oracle/filler.py`` from (name, shape, seed).  The only shim is an in-memory
``timm.models.layers`` (DropPath identity, to_2tuple, trunc_normal_), because timm is not
installed here and the reference uses just those three names (swin_gnn.py:5).
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

# ---- timm shim (3 names) -----------------------------------------------------------------
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
from runner.mcmc_sampler.edm import EDMSampler as REDMSampler  # noqa: E402
from runner.objectives.edm import EDMObjectiveGenerator as RObj  # noqa: E402
from loss.rainbow_loss import RainbowLoss as RLoss  # noqa: E402
from utils.attribute_code import attribute_converter as r_attribute_converter  # noqa: E402

from oracle import filler  # noqa: E402
from tests.golden.cases import NET_CASES, BLOCK_CASES, ATTN_CASES, MERGE_CASES, EMBED_CASES, SAMPLER_CASES  # noqa: E402

torch.set_grad_enabled(False)
OUT = {}
META = {}


def fill(module, seed, regime):
    module.load_state_dict(filler.fill_state_dict(module.state_dict(), seed, regime), strict=True)
    return module


def emb_for(B, seed):
    return filler.uniform("emb", (B, 512), seed)


# ---- G1: WindowAttention ------------------------------------------------------------------
for name, c in ATTN_CASES.items():
    m = fill(R.WindowAttention(c["C"], (c["w"], c["w"]), c["heads"]), c["seed"], "stress").eval()
    n = c["w"] ** 2
    x = filler.uniform("x", (c["B_"], n, c["C"]), c["seed"])
    mask = None
    if c["mask_res"]:
        blk = R.SwinTransformerBlock(c["C"], 512, (c["mask_res"],) * 2, c["heads"], c["w"], c["w"] // 2)
        mask = blk.attn_mask
    OUT[f"attn/{name}"] = m(x, mask).numpy()

# ---- G2/G3: SwinTransformerBlock ----------------------------------------------------------
for name, c in BLOCK_CASES.items():
    m = R.SwinTransformerBlock(c["C"], 512, (c["res"],) * 2, c["heads"], c["window"], c["shift"]).eval()
    if m.attn_mask is not None:
        OUT[f"block/{name}/attn_mask"] = m.attn_mask.numpy()
    OUT[f"block/{name}/rel_index"] = m.attn.relative_position_index.numpy()
    fill(m, c["seed"], "stress")
    x = filler.uniform("x", (c["B"], c["res"] ** 2, c["C"]), c["seed"])
    OUT[f"block/{name}/out"] = m(x, emb_for(c["B"], c["seed"])).numpy()

# ---- G4: PatchMerging / PatchBreakup ------------------------------------------------------
for name, c in MERGE_CASES.items():
    m = fill(R.PatchMerging((c["res"],) * 2, c["C"]), c["seed"], "stress").eval()
    x = filler.uniform("x", (c["B"], c["res"] ** 2, c["C"]), c["seed"])
    OUT[f"merge/{name}"] = m(x).numpy()
    b = fill(R.PatchBreakup((c["res"] // 2,) * 2, 4 * c["C"]), c["seed"], "stress").eval()
    xb = filler.uniform("xb", (c["B"], (c["res"] // 2) ** 2, 4 * c["C"]), c["seed"])
    OUT[f"breakup/{name}"] = b(xb).numpy()

# ---- G5: PatchEmbed ------------------------------------------------------------------------
for name, c in EMBED_CASES.items():
    m = fill(R.PatchEmbed(c["N"], c["p"], c["in"], c["C"], 512, torch.nn.LayerNorm), c["seed"], "stress").eval()
    x = filler.uniform("x", (c["B"], c["in"], c["N"], c["N"]), c["seed"])
    OUT[f"embed/{name}"] = m(x, emb_for(c["B"], c["seed"]), None).numpy()

# ---- G6/G7: SwinGNN and Precond(SwinGNN) --------------------------------------------------
for name, c in NET_CASES.items():
    cfg = c["cfg"]
    net = R.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                    window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=cfg["self_condition"]).eval()
    pre = RPrecond("edm", net, cfg["self_condition"]).eval()
    META[f"keys/{name}"] = {k: [list(v.shape), str(v.dtype)] for k, v in pre.state_dict().items()}
    META[f"nparams/{name}"] = sum(p.numel() for p in pre.parameters())
    B, N = c["B"], cfg["img_size"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], c["seed"])
    for regime in c["regimes"]:
        fill(pre, c["seed"], regime)
        for si, sigma in enumerate(c["sigmas"]):
            x = filler.sym_adj("x", B, N, flags, c["seed"] + si, scale=float(np.sqrt(sigma ** 2 + 0.25)))
            sig = torch.full((B,), float(sigma)) * (1.0 + 0.1 * filler.uniform("sig", (B,), c["seed"])).abs()
            sc = filler.sym_adj("sc", B, N, flags, c["seed"] + 100 + si, scale=0.5)
            c_noise = sig.log() / 4
            OUT[f"net/{name}/{regime}/s{si}/F_nosc"] = net(x, flags, c_noise, None).numpy()
            OUT[f"net/{name}/{regime}/s{si}/F_sc"] = net(x, flags, c_noise, sc).numpy()
            for coin, u in (("coin1", 0.25), ("coin0", 0.75)):
                saved = np.random.rand
                np.random.rand = lambda u=u: u
                try:
                    OUT[f"net/{name}/{regime}/s{si}/D_{coin}"] = pre(x.clone(), flags, sig).numpy()
                finally:
                    np.random.rand = saved

# ---- G8: EDMSampler.sample -----------------------------------------------------------------
for name, c in SAMPLER_CASES.items():
    cfg = NET_CASES[c["net"]]["cfg"]
    net = R.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                    window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=cfg["self_condition"]).eval()
    pre = fill(RPrecond("edm", net, cfg["self_condition"]), c["seed"], c["regime"]).eval()
    smp = REDMSampler(num_steps=c["steps"], clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                      clip_samples_scope="x_0", self_condition=cfg["self_condition"], dev="cpu")
    OUT[f"sampler/{name}/sigma_steps"] = smp.sigma_steps.numpy()
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    gt = None
    if c["sanity"]:
        gt = filler.sym_adj("gt", c["B"], cfg["img_size"], flags, c["seed"]).sign()
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    final, interim = smp.sample(pre, flags, sanity_check_gt_adjs=gt, flag_interim_adjs=True, max_num_interim_adjs=5)
    OUT[f"sampler/{name}/final"] = final.numpy()
    OUT[f"sampler/{name}/interim"] = interim.numpy()
    q = r_attribute_converter(in_attr=final.clamp(min=-1.0, max=1.0), attr_flags=flags, in_encoding="ddpm",
                              out_encoding="int", num_attr_type=2, flag_nodes=True, flag_adjs=False,
                              flag_in_ddpm_range=True, flag_out_ddpm_range=False)
    OUT[f"sampler/{name}/thresholded"] = q.numpy().astype(np.float32)

# ---- seeded initialisation fingerprint (SwinGNN._init_weights under torch.manual_seed(1234)) ----------
for name in ("ego60", "grid192"):
    cfg = NET_CASES[name]["cfg"]
    torch.manual_seed(1234)
    net = R.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                    window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=cfg["self_condition"])
    sd = RPrecond("edm", net, cfg["self_condition"]).state_dict()
    keys = [k for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith("attn_mask")]
    OUT[f"init/{name}/sums"] = np.array([float(sd[k].double().sum()) for k in keys])
    OUT[f"init/{name}/abs_sums"] = np.array([float(sd[k].double().abs().sum()) for k in keys])
    META[f"init_keys/{name}"] = keys

# ---- objective + loss (a18, a19) -----------------------------------------------------------
B, N = 6, 24
flags = filler.node_flags("flags", B, N, 5, 20, 3)
clean = filler.sym_adj("clean", B, N, flags, 3).sign()
obj = RObj("edm", "edm", other_params=None, dev="cpu")
torch.manual_seed(5)
net_in, net_cond, net_tgt, (c_skip, c_out, c_in, c_noise, sigmas, weights) = obj.get_input_output(clean.clone(), flags)
OUT["objective/noisy"] = net_in.numpy()
OUT["objective/sigmas"] = sigmas.numpy()
OUT["objective/weights"] = weights.numpy()
pred = filler.sym_adj("pred", B, N, flags, 4)
loss = RLoss(regression_loss_weight=1.0, flag_reweight=False, objective="edm")
OUT["objective/loss"] = loss(net_pred=pred, net_target=net_tgt, net_cond=net_cond, adjs_perturbed=net_in,
                             adjs_gt=clean, node_flags=flags, loss_weight=weights, reduction="none").numpy()

np.savez_compressed(os.path.join(HERE, "golden.npz"), **OUT)
with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
    json.dump(META, f, indent=0, sort_keys=True)
print("wrote", len(OUT), "arrays;", sum(v.nbytes for v in OUT.values()) / 1e6, "MB raw")
