"""Multi-GPU plumbing of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in CPU tests).

Sampling shards as independent graphs: every rank samples its slice of the batch with its own seed and NO
collective is on the timed path (reference: ``DistributedSampler`` split in utils/dataloader.py:27-31 and
runner/sampler/sampler_utils.py:254-258, seed + rank in utils/arg_parser.py:254-255).  The only exchange is the
optional final gather of the samples (utils/dist_training.py:170-191).

Training has ONE exchange step, the gradient all-reduce (utils/dist_training.py:67-74 wraps the model in
``DistributedDataParallel``).  Two ways to get it here: wrap the drop-in ``Precond`` in stock DDP (its hooks see
ordinary parameter gradients), or -- the native route -- ``enable_gradient_sync(model)``: the HIP backward writes all
kernel-side gradients into ONE flat fp32 buffer from the tail forwards, so ``GradSync`` launches a few large
asynchronous RCCL all-reduces on slices of that buffer while the remaining backward kernels run (xGMI rings are
per-link bound: few, large messages), instead of DDP's bucket copies.  Use one of the two, not both.
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


class GradSync:
    """Bucketed asynchronous SUM all-reduce over slices of one flat gradient buffer that becomes ready from its tail
    towards its head (the order the backward pass produces it).  The caller pre-scales the loss gradient by
    1 / world_size, so the sum is the DDP average (gradients are linear in it)."""

    def __init__(self, group=None, bucket_bytes: int = 64 << 20):
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.handles = []
        self.sent_lo = None   # everything in [sent_lo, end) has been handed to the collective
        self.ready_lo = None
        self.flat = None

    @property
    def world_size(self):
        return dist.get_world_size(self.group)

    def begin(self, flat: torch.Tensor):
        self.flat = flat
        self.sent_lo = self.ready_lo = flat.numel()
        self.handles = []

    def ready_from(self, lo: int):
        """Declare flat[lo:] complete (kernels that write it are enqueued on the current stream)."""
        self.ready_lo = min(self.ready_lo, lo)
        if self.sent_lo - self.ready_lo >= self.bucket_elems:
            self._send(self.ready_lo)

    def _send(self, lo: int):
        if lo < self.sent_lo:
            self.handles.append(dist.all_reduce(self.flat[lo:self.sent_lo], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))
            self.sent_lo = lo

    def finish(self, extra=()):
        """Flush the head of the buffer, reduce the small `extra` tensors (one coalesced message) and wait."""
        self._send(0)
        extra = [t for t in extra if t is not None]
        if extra:
            buf = torch.cat([t.reshape(-1) for t in extra])
            self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        if extra:
            o = 0
            for t in extra:
                t.copy_(buf[o:o + t.numel()].view(t.shape))
                o += t.numel()
        self.handles = []
        return extra


def enable_gradient_sync(model, group=None, bucket_bytes: int = 64 << 20, min_world: int = 2):
    """Data-parallel training without the DDP wrapper: the HIP backward of `model` (a SwinGNN, or the `.model` of a
    Precond) averages its gradients over `group` while it runs.  Pass ``None`` world (not initialised) to disable.
    `min_world=1` keeps the collectives on for a one-rank group (tests: the RCCL calls are issued and are identities)."""
    net = getattr(model, "model", model)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= min_world:
        net._swg_grad_sync = GradSync(group, bucket_bytes)
    else:
        net._swg_grad_sync = None
    return model
