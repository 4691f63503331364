"""Time the full train step (objective -> Precond forward -> loss -> backward -> clip + Adam + 5 EMA) on one MI355X.
usage: bench_train.py [com|grid|dd] [batch] [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import prof

CFG = {"com": dict(N=24, p=3, depths=[4, 4, 6], heads=[3, 6, 12], w=24, n_lo=12, n_hi=20, B=80),
       "grid": dict(N=384, p=4, depths=[1, 1, 3, 1], heads=[3, 6, 12, 24], w=6, n_lo=361, n_hi=361, B=8),
       "dd": dict(N=512, p=4, depths=[1, 1, 3, 1], heads=[3, 6, 12, 24], w=8, n_lo=100, n_hi=500, B=4)}
name = sys.argv[1] if len(sys.argv) > 1 else "com"
c = CFG[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else c["B"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(1234); np.random.seed(1234)
dev = "cuda:0"
net = S.SwinGNN(img_size=c["N"], patch_size=c["p"], in_chans=1, embed_dim=96, depths=c["depths"], num_heads=c["heads"],
                window_size=c["w"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0., drop_path_rate=0., self_condition=True)
model = S.Precond("edm", net, True).to(dev).train()
opt = S.FusedAdamEMA(model.parameters(), lr=1e-4, max_norm=10.0, ema_betas=(0.9, 0.95, 0.99, 0.999, 0.9999))
obj = S.EDMObjectiveGenerator("edm", "edm", dev=dev)
loss_fn = S.RainbowLoss(1.0, False, "edm")
n = torch.randint(c["n_lo"], c["n_hi"] + 1, (B,))
flags = (torch.arange(c["N"])[None, :] < n[:, None]).float().to(dev)
a = (torch.rand(B, c["N"], c["N"], device=dev) < 0.3).float() * 2 - 1
a = torch.triu(a, 1); clean = (a + a.transpose(1, 2)) * flags[:, :, None] * flags[:, None, :]

def step():
    opt.zero_grad(set_to_none=True)
    x, sig, tgt, (cs, co, ci, cn, sigmas, w) = obj.get_input_output(clean, flags)
    D = model(x=x, node_flags=flags, sigmas=sig)
    loss = loss_fn(D, tgt, sig, x, clean, flags, loss_weight=w, reduction='none').mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(3): l = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps): l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"{name} B={B}: {dt * 1e3:.2f} ms/step  {B / dt:.1f} graphs/s  loss {float(l.detach()):.4f}  params {sum(p.numel() for p in model.parameters())/1e6:.1f}M")
with prof.profiled():
    for _ in range(3): step()
    torch.cuda.synchronize()
    rows = prof.read()
tot = sum(r["ms"] for r in rows.values())
print(f"kernel time per step {tot / 3:.2f} ms")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print(f"  {k:22s} {r['ms'] / 3:8.3f} ms  {100 * r['ms'] / tot:5.1f}%  launches {r['launches'] // 3}")
