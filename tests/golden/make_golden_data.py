"""Goldens of the host-side data formats from the REFERENCE's own helpers: dataset tensorisation
(utils/dataloader.py:219-247 = pad_adjs + flags + attribute_converter int -> ddpm) on synthetic graphs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_data.py     (build container only)
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

from utils.graph_utils import pad_adjs                              # noqa: E402  (reference)
from utils.attribute_code import attribute_converter                # noqa: E402

from tests.helpers import synthetic_graph_list                      # noqa: E402

N = 20
graphs = synthetic_graph_list()
adjs_np = np.asarray([pad_adjs(g, node_number=N) for g in graphs])
adjs = torch.tensor(adjs_np, dtype=torch.float32)
flags = adjs.sum(-1).gt(1e-5).to(dtype=torch.float32)
scaled = attribute_converter(in_attr=adjs, attr_flags=flags, in_encoding='int', out_encoding='ddpm', num_attr_type=2,
                             flag_nodes=True, flag_adjs=False, flag_in_ddpm_range=False, flag_out_ddpm_range=True)
np.savez_compressed(os.path.join(HERE, "golden_data.npz"), adjs=scaled.numpy(), flags=flags.numpy())
print(scaled.shape, flags.sum(-1))
