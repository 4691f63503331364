"""Golden vectors of full 256-step EDMSampler.sample runs by the REFERENCE's own modules on CPU (same recipe as
make_golden.py G8, kept in a file of its own because the grid geometry takes about a minute of CPU).

Run in the build container only (needs /root/reference; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sampler_long.py

Committed: golden_sampler_long.npz (final samples, thresholded samples, a few interim states).  Weights, flags and noise
are regenerated on both sides from (name, shape, seed) / the seeded CPU generator.

`--only full` / `--only long` restricts the run.  The `full` set (SAMPLER_FULL_CASES -> golden_sampler_full.npz) holds
few-step trajectories at the FULL bench geometries (grid N=384 and DD N=512: 8 Heun steps, B = 1; final sample
and thresholded sample only).
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

from model.swin_gnn import swin_gnn as R          # noqa: E402  (reference)
from model.precond.precond import Precond as RPrecond  # noqa: E402
from runner.mcmc_sampler.edm import EDMSampler as REDMSampler  # noqa: E402
from utils.attribute_code import attribute_converter as r_attribute_converter  # noqa: E402

from oracle import filler  # noqa: E402
from tests.golden.cases import NET_CASES, SAMPLER_FULL_CASES, SAMPLER_LONG_CASES  # noqa: E402

torch.set_grad_enabled(False)
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "both"
OUT, FULL = {}, {}
for name, c in list(SAMPLER_LONG_CASES.items()) + list(SAMPLER_FULL_CASES.items()):
    full = name in SAMPLER_FULL_CASES
    if (full and ONLY == "long") or (not full and ONLY == "full"):
        continue
    cfg = c["cfg"] if full else NET_CASES[c["net"]]["cfg"]
    net = R.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                    window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=cfg["self_condition"]).eval()
    pre = RPrecond("edm", net, cfg["self_condition"])
    pre.load_state_dict(filler.fill_state_dict(pre.state_dict(), c["seed"], c["regime"]), strict=True)
    pre.eval()
    smp = REDMSampler(num_steps=c["steps"], clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                      clip_samples_scope="x_0", self_condition=cfg["self_condition"], dev="cpu")
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    final, interim = smp.sample(pre, flags, flag_interim_adjs=True, max_num_interim_adjs=5)
    dst = FULL if full else OUT
    dst[f"sampler/{name}/final"] = final.numpy()
    if not full:
        dst[f"sampler/{name}/interim"] = interim.numpy()
    q = r_attribute_converter(in_attr=final.clamp(min=-1.0, max=1.0), attr_flags=flags, in_encoding="ddpm",
                              out_encoding="int", num_attr_type=2, flag_nodes=True, flag_adjs=False,
                              flag_in_ddpm_range=True, flag_out_ddpm_range=False)
    dst[f"sampler/{name}/thresholded"] = q.numpy().astype(np.uint8 if full else np.float32)
    print(name, "done", flush=True)

for fname, arrays in (("golden_sampler_long.npz", OUT), ("golden_sampler_full.npz", FULL)):
    if arrays:
        np.savez_compressed(os.path.join(HERE, fname), **arrays)
        print("wrote", fname, len(arrays), "arrays;", sum(v.nbytes for v in arrays.values()) / 1e6, "MB raw")
