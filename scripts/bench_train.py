"""Time the full train step (objective -> Precond forward -> loss -> backward -> clip + Adam + 5 EMA) on MI355X.

    python scripts/bench_train.py [com|grid|dd] [batch] [steps]                       one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        scripts/bench_train.py grid 80 10 --ddp stock|native                          N GPUs, one rank per GPU (RCCL)

--ddp stock : the drop-in module wrapped in torch's DistributedDataParallel, as utils/dist_training.py:67-74 does
--ddp native: parallel.enable_gradient_sync -- the HIP backward's flat gradient buffer is all-reduced in a few large
              asynchronous RCCL messages while the remaining backward kernels run (no DDP wrapper, no bucket copies)
`batch` is the PER-RANK batch (weak scaling: the reference's per-GPU batch is batch_size / world, dataloader.py:27-31).
Rank 0 prints one JSON line in DDP mode (ms per step = max over ranks, graphs/s = all ranks' graphs / that time).
SWG_BENCH_REHEARSAL=1: ranks share the visible GPU(s) and meet over gloo (code-path check on a one-GPU box, not a
scaling number)."""
import argparse, faulthandler, json, os, sys, time
import numpy as np, torch
if os.environ.get("SWG_BENCH_WATCHDOG"):   # dump every thread's Python stack if the script is still running after N s
    faulthandler.dump_traceback_later(int(os.environ["SWG_BENCH_WATCHDOG"]), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import prof

CFG = {"com": dict(N=24, p=3, depths=[4, 4, 6], heads=[3, 6, 12], w=24, n_lo=12, n_hi=20, B=80),
       "grid": dict(N=384, p=4, depths=[1, 1, 3, 1], heads=[3, 6, 12, 24], w=6, n_lo=361, n_hi=361, B=8),
       "dd": dict(N=512, p=4, depths=[1, 1, 3, 1], heads=[3, 6, 12, 24], w=8, n_lo=100, n_hi=500, B=4)}
ap = argparse.ArgumentParser()
ap.add_argument("name", nargs="?", default="com", choices=list(CFG))
ap.add_argument("batch", nargs="?", type=int, default=None)
ap.add_argument("steps", nargs="?", type=int, default=10)
ap.add_argument("--ddp", choices=("stock", "native"), default=None)
args = ap.parse_args()
name, c = args.name, CFG[args.name]
B = args.batch or c["B"]
steps = args.steps
rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
rehearsal = os.environ.get("SWG_BENCH_REHEARSAL") == "1"
dev_index = local_rank % max(torch.cuda.device_count(), 1) if rehearsal else local_rank
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
dist = None
if world > 1 or args.ddp:
    # world 1 with --ddp: a ONE-rank RCCL group (wrapper / hook / collective-call overhead on one GPU; every
    # collective is an identity)
    import torch.distributed as dist
    assert args.ddp, "several ranks need --ddp stock|native"
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("gloo" if rehearsal else "nccl", **({} if rehearsal else {"device_id": dev}))
torch.manual_seed(1234)           # identical initial weights on every rank (DDP broadcasts them anyway)
np.random.seed(1234 + rank)
net = S.SwinGNN(img_size=c["N"], patch_size=c["p"], in_chans=1, embed_dim=96, depths=c["depths"], num_heads=c["heads"],
                window_size=c["w"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0., drop_path_rate=0., self_condition=True)
model = S.Precond("edm", net, True).to(dev).train()
call = model
if args.ddp == "stock":
    call = torch.nn.parallel.DistributedDataParallel(model, device_ids=None if rehearsal else [dev_index])
elif args.ddp == "native":
    from swingnn_amd import parallel
    parallel.enable_gradient_sync(model, min_world=1)
opt = S.FusedAdamEMA(model.parameters(), lr=1e-4, max_norm=10.0, ema_betas=(0.9, 0.95, 0.99, 0.999, 0.9999))
obj = S.EDMObjectiveGenerator("edm", "edm", dev=dev)
loss_fn = S.RainbowLoss(1.0, False, "edm")
torch.manual_seed(1234 + rank)    # per-rank data and noise (utils/arg_parser.py:254-255)
n = torch.randint(c["n_lo"], c["n_hi"] + 1, (B,))
flags = (torch.arange(c["N"])[None, :] < n[:, None]).float().to(dev)
a = (torch.rand(B, c["N"], c["N"], device=dev) < 0.3).float() * 2 - 1
a = torch.triu(a, 1); clean = (a + a.transpose(1, 2)) * flags[:, :, None] * flags[:, None, :]

def step():
    opt.zero_grad(set_to_none=True)
    x, sig, tgt, (cs, co, ci, cn, sigmas, w) = obj.get_input_output(clean, flags)
    D = call(x=x, node_flags=flags, sigmas=sig)
    loss = loss_fn(D, tgt, sig, x, clean, flags, loss_weight=w, reduction='none').mean()
    loss.backward()
    opt.step()
    return loss

def barrier():
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()

for _ in range(3): l = step()
barrier()
t0 = time.time()
for _ in range(steps): l = step()
barrier()
dt = (time.time() - t0) / steps
if dist is not None:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # parameters must be identical on every rank after synchronised steps
    p0 = torch.cat([p.detach().reshape(-1)[:64] for p in model.parameters()])
    ref = p0.clone()
    dist.broadcast(ref, 0)
    same = bool(torch.equal(p0, ref))
    every = [None] * world
    dist.all_gather_object(every, same)
    if rank == 0:
        print(json.dumps(dict(workload=f"{name} dim-96 train step", ddp=args.ddp, backend=dist.get_backend(),
                              dist_world_size=dist.get_world_size(), rccl=not rehearsal, per_rank_batch=B,
                              global_batch=B * world, ms_per_step=dt * 1e3, graphs_per_s=B * world / dt,
                              loss_rank0=float(l.detach()), parameters_identical_across_ranks=all(every),
                              rehearsal=rehearsal)))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
print(f"{name} B={B}: {dt * 1e3:.2f} ms/step  {B / dt:.1f} graphs/s  loss {float(l.detach()):.4f}  params {sum(p.numel() for p in model.parameters())/1e6:.1f}M")
with prof.profiled():
    for _ in range(3): step()
    torch.cuda.synchronize()
    rows = prof.read()
tot = sum(r["ms"] for r in rows.values())
print(f"kernel time per step {tot / 3:.2f} ms")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print(f"  {k:22s} {r['ms'] / 3:8.3f} ms  {100 * r['ms'] / tot:5.1f}%  launches {r['launches'] // 3}")
