"""Multi-GPU plumbing of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in CPU tests).

Sampling shards as independent graphs: every rank samples its slice of the batch with its own seed and NO
collective is on the timed path (reference: ``DistributedSampler`` split in utils/dataloader.py:27-31 and
runner/sampler/sampler_utils.py:254-258, seed + rank in utils/arg_parser.py:254-255).  The only exchange is the
optional final gather of the samples (utils/dist_training.py:170-191).  Training uses stock
``DistributedDataParallel`` around the drop-in module (utils/dist_training.py:67-74): the gradient all-reduce is
RCCL's, nothing here re-implements it.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def rank_seed(seed: int, rank: int) -> int:
    """Per-rank seed rule of the reference (utils/arg_parser.py:254-255)."""
    return seed + rank


def shard_rows(t: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """Rows of ``t`` that ``rank`` owns under the DistributedSampler(shuffle=False) rule: indices
    rank, rank + W, rank + 2W, ... after padding the index list to a multiple of W by wrapping around."""
    n = t.shape[0]
    total = (n + world_size - 1) // world_size * world_size
    idx = torch.arange(total) % n
    return t[idx[rank:total:world_size]]


def gather_rows(t: torch.Tensor, cat_dim: int = 0) -> torch.Tensor:
    """All-gather equally-shaped per-rank tensors and concatenate along ``cat_dim`` (rank-major), the contract of
    gather_tensors (utils/dist_training.py:170-191) without its cat_dim=1 layout bug."""
    rank, world_size = world()
    if world_size == 1:
        return t
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(world_size)]
    dist.all_gather(parts, t)
    return torch.cat(parts, dim=cat_dim)


def sample_sharded(sampler, model, node_flags: torch.Tensor, seed: int, gather: bool = True, **kw):
    """Draw samples for the rows of ``node_flags`` this rank owns (seeded seed + rank) and optionally gather.
    Returns (samples [B_local or B_total, N, N] on CPU, local node_flags)."""
    import numpy as np
    rank, world_size = world()
    local = shard_rows(node_flags, rank, world_size)
    torch.manual_seed(rank_seed(seed, rank))
    np.random.seed(rank_seed(seed, rank))
    out = sampler.sample(model, local, **kw)
    if gather and world_size > 1:
        dev = local.device if local.device.type == "cuda" else torch.device("cpu")
        out = gather_rows(out.to(dev)).cpu()
    return out, local
