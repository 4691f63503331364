import json, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swingnn_amd as S
from oracle import filler, swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NET_CASES
from tests.test_gpu_parity import build_precond
torch.set_grad_enabled(False)
meta = json.load(open('tests/golden/golden_meta.json'))
c = NET_CASES['ego60']; cfg = c['cfg']
sd = H.net_state(meta["keys/ego60"], 61, 'init'); net = O._strip(sd)
pre = build_precond(S, 'ego60', meta, 'init')
pre.load_state_dict({k: v for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith('attn_mask')}, strict=False); pre = pre.cuda()
B, N = 3, 24
flags = filler.node_flags("flags", B, N, 4, 18, 61)
torch.manual_seed(0)
for sigma in [1.0, 40.0, 80.0, 113.0, 200.0]:
    for kind in ['sym', 'folded']:
        raw = torch.randn(B, N, N)
        x = (O.gen_init_sample(flags, raw) if kind == 'folded' else O.mask_adjs(O.sym_noise_from_raw(raw), flags)) * sigma
        sig = torch.full((B,), sigma)
        cs, co, ci, cn = O.edm_coefficients(sig)
        Fr = O.swingnn_forward(net, cfg, ci.view(-1,1,1)*x, flags, cn, None)
        Fg = pre.model((ci.view(-1,1,1)*x).cuda(), flags.cuda(), cn.cuda(), None).cpu()
        Dr = O.precond_forward(sd, cfg, x, flags, sig, False)
        np.random.rand = lambda: 0.9
        Dg = pre(x.cuda(), flags.cuda(), sig.cuda()).cpu()
        Dg_u = pre(x.cuda(), flags.cuda(), torch.tensor([sigma]).cuda().expand(B)).cpu()
        print(f"sigma {sigma} {kind}: F err {float((Fr-Fg).abs().max()):.3g} (scale {float(Fr.abs().max()):.3g})  D err {float((Dr-Dg).abs().max()):.3g} uniform {float((Dr-Dg_u).abs().max()):.3g} (scale {float(Dr.abs().max()):.3g})")
