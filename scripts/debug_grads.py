"""Per-parameter gradient error of the HIP train step vs autograd through the CPU oracle (run on the GPU box)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NET_CASES
oracle_grads = H.oracle_grads
import swingnn_amd as S
from tests.test_gpu_parity import build_precond
torch.set_grad_enabled(True)
name = sys.argv[1] if len(sys.argv) > 1 else "com96"
regime = sys.argv[2] if len(sys.argv) > 2 else "stress"
coin = int(sys.argv[3]) if len(sys.argv) > 3 else 1
meta = json.load(open(os.path.join(ROOT, "tests/golden/golden_meta.json")))
c = NET_CASES[name]
sd = H.net_state(meta[f"keys/{name}"], c["seed"], regime)
flags, x, sig, _ = H.net_inputs(c, 0)
clean = O.mask_adjs(torch.sign(x), flags)
w = (sig ** 2 + 0.25) / (sig * 0.5) ** 2
ref_loss, ref = oracle_grads(sd, c["cfg"], x, flags, sig, w, clean, bool(coin))
pre = build_precond(S, name, meta, regime).train()
np.random.rand = lambda: 0.25 if coin else 0.75
D = pre(x=x.cuda(), node_flags=flags.cuda(), sigmas=sig.cuda())
loss = S.RainbowLoss(1.0, False, "edm")(D, clean.cuda(), sig.cuda(), x.cuda(), clean.cuda(), flags.cuda(), loss_weight=w.cuda(), reduction='none')
loss.mean().backward()
print("loss", loss.detach().cpu(), ref_loss)
total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
for k, p in pre.named_parameters():
    g = p.grad.float().cpu(); r = ref[k]
    nr = float(r.norm()); err = float((g - r).norm())
    cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-30))
    print(f"{k:60s} ref {nr:10.4g} got {float(g.norm()):10.4g} rel {err / max(nr, 1e-30):8.4f} cos {cos:7.4f}")
