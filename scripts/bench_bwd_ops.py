"""Time of every op-level entry point of one train step (forward and backward), per call site, with a device sync
around each call (serialised: the sum is an upper bound of the step).  usage: bench_bwd_ops.py [grid|com|dd] [batch]"""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import train as T

CFG = {"com": dict(N=24, p=3, depths=[4, 4, 6], heads=[3, 6, 12], w=24, n_lo=12, n_hi=20, B=80),
       "grid": dict(N=384, p=4, depths=[1, 1, 3, 1], heads=[3, 6, 12, 24], w=6, n_lo=361, n_hi=361, B=80),
       "dd": dict(N=512, p=4, depths=[1, 1, 3, 1], heads=[3, 6, 12, 24], w=8, n_lo=100, n_hi=500, B=4)}
name = sys.argv[1] if len(sys.argv) > 1 else "grid"
c = CFG[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else c["B"]
torch.manual_seed(1234); np.random.seed(1234)
dev = "cuda:0"
net = S.SwinGNN(img_size=c["N"], patch_size=c["p"], in_chans=1, embed_dim=96, depths=c["depths"], num_heads=c["heads"],
                window_size=c["w"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0., drop_path_rate=0., self_condition=True)
model = S.Precond("edm", net, True).to(dev).train()
obj = S.EDMObjectiveGenerator("edm", "edm", dev=dev)
loss_fn = S.RainbowLoss(1.0, False, "edm")
n = torch.randint(c["n_lo"], c["n_hi"] + 1, (B,))
flags = (torch.arange(c["N"])[None, :] < n[:, None]).float().to(dev)
a = (torch.rand(B, c["N"], c["N"], device=dev) < 0.3).float() * 2 - 1
a = torch.triu(a, 1); clean = (a + a.transpose(1, 2)) * flags[:, :, None] * flags[:, None, :]

def step():
    model.zero_grad(set_to_none=True)
    x, sig, tgt, (cs, co, ci, cn, sigmas, w) = obj.get_input_output(clean, flags)
    D = model(x=x, node_flags=flags, sigmas=sig)
    loss_fn(D, tgt, sig, x, clean, flags, loss_weight=w, reduction='none').mean().backward()

for _ in range(2): step()
torch.cuda.synchronize()
tally = collections.OrderedDict()
orig = T._run
def timed(fn, what, *args):
    # key: entry point + channel count of the block / merge / breakup descriptor when there is one
    key = what
    try:
        d = args[0]._obj if hasattr(args[0], "_obj") else None
        if d is not None and hasattr(d, "C"): key += f" C={d.C}"
        elif d is not None and hasattr(d, "Cin"): key += f" Cin={d.Cin}"
    except Exception:
        pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orig(fn, what, *args)
    torch.cuda.synchronize()
    e = tally.setdefault(key, [0.0, 0]); e[0] += time.perf_counter() - t0; e[1] += 1
T._run = timed
reps = 3
for _ in range(reps): step()
T._run = orig
tot = sum(v[0] for v in tally.values()) / reps
print(f"{name} B={B}: op-level sum {tot * 1e3:.2f} ms per step")
for k, (t, nc) in sorted(tally.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:34s} {t / reps * 1e3:8.3f} ms  {nc // reps:3d} calls  {100 * t / reps / tot:5.1f}%")
