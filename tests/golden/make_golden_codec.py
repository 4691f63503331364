"""Fixtures of the attribute codecs from the REFERENCE's own utils/attribute_code.py (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_codec.py     -> tests/golden/golden_codec.npz

Inputs are regenerated on both sides by oracle/filler.py; only expected outputs are stored.  The post-sampling decode
(runner/sampler/sampler_node_adj.py:178-200) consists of three lines around ``attribute_converter`` inside a closure of
``mol_go_sampling``; they are restated in ``ref_decode`` below (clamp, binarise at 0, mask) and everything else is the
reference's function.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from utils.attribute_code import attribute_converter as r_conv  # noqa: E402  (reference)
from utils.graph_utils import mask_adjs as r_mask_adjs, mask_nodes as r_mask_nodes  # noqa: E402

from oracle import filler  # noqa: E402
from tests.golden.cases import CODEC_CASES  # noqa: E402


def ref_decode(x, flags, enc, k, is_adj):
    x = x.clamp(-1.0, 1.0)
    if enc in ("bits", "one_hot"):
        x = torch.where(x > 0.0, torch.ones_like(x), -torch.ones_like(x))
        x = (r_mask_adjs if is_adj else r_mask_nodes)(x, flags)
    return r_conv(in_attr=x, attr_flags=flags, in_encoding=enc, out_encoding="int", num_attr_type=k, flag_nodes=True,
                  flag_adjs=False, flag_in_ddpm_range=True, flag_out_ddpm_range=False, flag_clamp_int=enc == "bits")


OUT = {}
for name, c in CODEC_CASES.items():
    B, N, k, seed = c["B"], c["N"], c["k"], c["seed"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], seed)
    for is_adj in (False, True):
        tag = f"{name}/{'adj' if is_adj else 'node'}"
        shape = (B, N, N) if is_adj else (B, N)
        # integer attributes in [0, k): every encoding of them, and back
        ints = ((filler.uniform(tag + "/int", shape, seed) + 1.0) * 0.5 * k).floor().clamp(0, k - 1)
        ints = (r_mask_adjs if is_adj else r_mask_nodes)(ints, flags)
        for enc in ("ddpm", "one_hot", "bits"):
            e = r_conv(ints, flags, "int", enc, k, flag_nodes=not is_adj, flag_adjs=is_adj, flag_in_ddpm_range=False,
                       flag_out_ddpm_range=True)
            OUT[f"{tag}/enc_{enc}"] = e.numpy().astype(np.float32)
            back = r_conv(e, flags, enc, "int", k, flag_nodes=not is_adj, flag_adjs=is_adj, flag_in_ddpm_range=True,
                          flag_out_ddpm_range=True)
            assert (back == ints).all()
        # raw continuous "samples" through the sampler's decode
        for enc in ("ddpm", "one_hot", "bits"):
            C = {"ddpm": 1, "one_hot": k, "bits": int(np.ceil(np.log2(k)))}[enc]
            if enc == "ddpm":
                raw = filler.normalish(tag + "/raw_ddpm", shape, seed) * 0.9
            elif is_adj:
                raw = filler.normalish(tag + f"/raw_{enc}", (B, C, N, N), seed) * 0.9
                raw = raw + raw.transpose(-1, -2)
            else:
                raw = filler.normalish(tag + f"/raw_{enc}", (B, N, C), seed) * 0.9
            OUT[f"{tag}/dec_{enc}"] = ref_decode(raw, flags, enc, k, is_adj).numpy().astype(np.float32)

np.savez_compressed(os.path.join(HERE, "golden_codec.npz"), **OUT)
print("wrote", len(OUT), "arrays;", sum(v.nbytes for v in OUT.values()) / 1e6, "MB raw")
