"""MI355X-native SwinGNN denoiser hot path: drop-in ``SwinGNN``, ``Precond`` and ``EDMSampler`` whose
arithmetic runs in hand-written gfx950 HIP kernels (libswingnn_hip.so, include/swingnn_hip.h).

Importing this package loads the shared library and raises ImportError if it is missing:
there is no CPU or PyTorch fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)
from .swin_gnn import (SwinGNN, SwinTransformerBlock, WindowAttention, PatchMerging, PatchBreakup, PatchEmbed,
                       BasicLayer, Mlp, PositionalEmbedding, NodeAdjSwinGNN)
from .precond import Precond, NodeAdjPrecond
from .edm_sampler import EDMSampler, GeneralSampler, NodeAdjEDMSampler
from .objective import EDMObjectiveGenerator, RainbowLoss, NodeAdjEDMObjectiveGenerator, NodeAdjRainbowLoss
from .optim import FusedAdamEMA
from .io import load_model, adjs_to_dataset_tensors, quantise_samples
from . import codec, graph_stats  # noqa: F401  (post-sampling: attribute codecs, graph statistics / MMD on the device)

__all__ = ["SwinGNN", "SwinTransformerBlock", "WindowAttention", "PatchMerging", "PatchBreakup", "PatchEmbed",
           "BasicLayer", "Mlp", "PositionalEmbedding", "Precond", "EDMSampler", "GeneralSampler", "EDMObjectiveGenerator", "RainbowLoss",
           "FusedAdamEMA", "NodeAdjSwinGNN", "NodeAdjPrecond", "NodeAdjEDMSampler", "NodeAdjEDMObjectiveGenerator", "NodeAdjRainbowLoss", "load_model", "adjs_to_dataset_tensors", "quantise_samples"]
