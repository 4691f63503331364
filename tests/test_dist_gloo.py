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
