"""Time the molecule train step (QM9 geometry: N=16, patch 1, window 4, one-hot 4+4 channels or ddpm 1+1) on one MI355X:
objective -> NodeAdjPrecond forward -> losses -> backward -> clip + Adam + 5 EMA.
usage: bench_train_qm9.py [onehot|ddpm] [dim] [batch] [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import prof

enc = sys.argv[1] if len(sys.argv) > 1 else "onehot"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
Ca = Cn = 4 if enc == "onehot" else 1
N = 16
torch.manual_seed(1234); np.random.seed(1234)
dev = "cuda:0"
net = S.NodeAdjSwinGNN(img_size=N, patch_size=1, in_chans=2 * Cn + Ca, embed_dim=dim, depths=[1, 1, 3, 1],
                       num_heads=[3, 6, 12, 24], window_size=4, mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                       drop_path_rate=0., self_condition=True, symmetric_noise=True, out_chans_adj=Ca, out_chans_node=Cn)
model = S.NodeAdjPrecond("edm", net, True, symmetric_noise=True).to(dev).train()
opt = S.FusedAdamEMA(model.parameters(), lr=1e-4, max_norm=10.0, ema_betas=(0.9, 0.95, 0.99, 0.999, 0.9999))
obj = S.NodeAdjEDMObjectiveGenerator("edm", "edm", dev=dev)
loss_fn = S.NodeAdjRainbowLoss(1.0, 1.0, "edm")
n = torch.randint(3, 10, (B,))
flags = (torch.arange(N)[None, :] < n[:, None]).float().to(dev)
shape_a = (B, N, N) if Ca == 1 else (B, Ca, N, N)
a = (torch.rand(shape_a, device=dev) < 0.3).float() * 2 - 1
a = torch.triu(a, 1); a = a + a.transpose(-1, -2)
fm = flags[:, :, None] * flags[:, None, :]
clean_a = a * (fm if Ca == 1 else fm[:, None])
x = (torch.rand((B, N) if Cn == 1 else (B, N, Cn), device=dev) < 0.5).float() * 2 - 1
clean_x = x * (flags if Cn == 1 else flags[:, :, None])

def step():
    opt.zero_grad(set_to_none=True)
    na, nx, sig, ta, tx, (cs, co, ci, cn, sigmas, w) = obj.get_input_output(clean_a, clean_x, flags)
    Da, Dn = model(adjs=na, nodes=nx, node_flags=flags, sigmas=sig)
    la, ln = loss_fn(Da, Dn, ta, tx, sig, na, clean_a, nx, clean_x, flags, loss_weight=w, reduction='none')
    loss = la.mean() + ln.mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(3): l = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps): l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"qm9 {enc} dim-{dim} B={B}: {dt * 1e3:.2f} ms/step  {B / dt:.1f} graphs/s  loss {float(l.detach()):.4f}  params {sum(p.numel() for p in model.parameters())/1e6:.1f}M")
with prof.profiled():
    for _ in range(3): step()
    torch.cuda.synchronize()
    rows = prof.read()
tot = sum(r["ms"] for r in rows.values())
print(f"kernel time per step {tot / 3:.2f} ms")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print(f"  {k:22s} {r['ms'] / 3:8.3f} ms  {100 * r['ms'] / tot:5.1f}%  launches {r['launches'] // 3}")
