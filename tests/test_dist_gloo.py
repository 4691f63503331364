"""World-size-2 gloo test of the N > 1 sampling path (swingnn_amd/parallel.py): batch sharding, per-rank seeds,
no collective on the sampling path, final gather.  The compute under the sharding is the CPU oracle (the HIP
sampler needs a GPU); what is checked is that the gathered result equals running each shard alone."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import filler
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NET_CASES

STEPS, SEED, B = 3, 77, 5  # odd batch: exercises the wrap-around padding of the shard rule


class OracleSampler:
    """EDMSampler-shaped object whose arithmetic is the CPU oracle."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def sample(self, model, node_flags, **kw):
        den = lambda x, fl, sig, sc: O.precond_forward(self.sd, self.cfg, x, fl, sig, np.random.rand() < 0.5)
        return O.edm_sample(den, node_flags, num_steps=STEPS, self_condition=True)


def _setup():
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(root, "tests", "golden", "golden_meta.json")))
    cfg = NET_CASES["ego60"]["cfg"]
    sd = H.net_state(meta["keys/ego60"], 5, "init")
    flags = filler.node_flags("flags", B, cfg["img_size"], 4, 18, 9)
    return OracleSampler(sd, cfg), flags


def _worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from swingnn_amd import parallel as P
    with torch.no_grad():
        smp, flags = _setup()
        out, local = P.sample_sharded(smp, None, flags, SEED, gather=True)
    if rank == 0:
        q.put((out.numpy(), local.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampling_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, local0 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # reference: each shard alone, same seeds
    from swingnn_amd import parallel as P
    smp, flags = _setup()
    expect = []
    with torch.no_grad():
        for rank in range(2):
            local = P.shard_rows(flags, rank, 2)
            torch.manual_seed(P.rank_seed(SEED, rank))
            np.random.seed(P.rank_seed(SEED, rank))
            expect.append(smp.sample(None, local))
    assert np.array_equal(local0, P.shard_rows(flags, 0, 2).numpy())
    assert P.shard_rows(flags, 0, 2).shape[0] == P.shard_rows(flags, 1, 2).shape[0] == 3  # 5 rows -> 3 + 3 (wrap)
    assert np.array_equal(gathered, torch.cat(expect).numpy())
    assert gathered.shape == (6, 24, 24)


def test_shard_rule_matches_distributed_sampler():
    from torch.utils.data import DistributedSampler, TensorDataset
    from swingnn_amd import parallel as P
    t = torch.arange(11)
    for world_size in (2, 4, 8):
        for rank in range(world_size):
            ref = list(DistributedSampler(TensorDataset(t), num_replicas=world_size, rank=rank, shuffle=False))
            assert P.shard_rows(t, rank, world_size).tolist() == ref


# ---- training: the one exchange step (gradient all-reduce), native route of swingnn_amd/parallel.py ----
def _train_setup():
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(root, "tests", "golden", "golden_meta.json")))
    c = NET_CASES["ego60"]
    sd = H.net_state(meta["keys/ego60"], c["seed"], "init")
    return c, sd, H.train_case_inputs(c, 1)


def _flat_grads(sd, grads):
    keys = [k for k, v in sd.items() if torch.is_floating_point(v) and k in grads]
    return keys, torch.cat([grads[k].reshape(-1) for k in keys])


def _train_worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from swingnn_amd import parallel as P
    c, sd, (flags, x, sig, w, clean) = _train_setup()
    sl = slice(rank, None, world_size)  # this rank's rows
    _, grads = H.oracle_grads(sd, c["cfg"], x[sl], flags[sl], sig[sl], w[sl], clean[sl], False)
    keys, flat = _flat_grads(sd, grads)
    n_extra = 3  # the last tensors play the part of the small torch-side gradients
    extra = [grads[k].clone() / world_size for k in keys[-n_extra:]]
    body = flat[:sum(grads[k].numel() for k in keys[:-n_extra])].clone() / world_size  # loss gradient pre-scaled by 1/W
    sync = P.GradSync(None, bucket_bytes=256 << 10)
    sync.begin(body)
    n = body.numel()
    for lo in (n - 1000, n - 70000, n // 2, n // 2 - 5, n // 7):  # the backward hands over its tail first
        sync.ready_from(lo)
    extra = sync.finish(extra)
    if rank == 0:
        q.put(torch.cat([body] + [t.reshape(-1) for t in extra]).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_sync_two_ranks_equals_full_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c, sd, (flags, x, sig, w, clean) = _train_setup()
    _, grads = H.oracle_grads(sd, c["cfg"], x, flags, sig, w, clean, False)  # mean over the whole batch
    _, ref = _flat_grads(sd, grads)
    ref = ref.numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
