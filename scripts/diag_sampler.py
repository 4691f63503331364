"""Diagnostic: trace every denoiser call of the HIP sampler against the CPU oracle trajectory."""
import json, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swingnn_amd as S
from oracle import filler, swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import NET_CASES, SAMPLER_CASES
from tests.test_gpu_parity import build_precond
torch.set_grad_enabled(False)
meta = json.load(open('tests/golden/golden_meta.json'))
name = sys.argv[1] if len(sys.argv) > 1 else 'ego60_8'
c = SAMPLER_CASES[name]; nc = NET_CASES[c['net']]; cfg = nc['cfg']
sd = H.net_state(meta[f"keys/{c['net']}"], c['seed'], c['regime'])
flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
rec_o = []
torch.manual_seed(c['seed']); np.random.seed(c['seed'])
def den(x, fl, sig, sc):
    coin = np.random.rand() < 0.5
    D = O.precond_forward(sd, cfg, x, fl, sig, coin)
    rec_o.append((x.clone(), float(sig[0]), coin, D.clone()))
    return D
ref = O.edm_sample(den, flags, num_steps=c['steps'], self_condition=True)
pre = build_precond(S, c['net'], meta, c['regime'], seed=c['seed'])
smp = S.EDMSampler(num_steps=c['steps'], clip_samples=False, clip_samples_min=-1., clip_samples_max=1., clip_samples_scope='x_0', self_condition=True, dev='cuda:0')
rec_g = []
class Wrap(torch.nn.Module):
    def __init__(s, m): super().__init__(); s.m = m
    def forward(s, x, fl, sig, sc):
        D = s.m(x, fl, sig, sc); rec_g.append((x.cpu().clone(), float(sig[0]), D.cpu().clone())); return D
    def assert_symmetric(s): s.m.assert_symmetric()
    @staticmethod
    def round_sigma(x): return torch.as_tensor(x)
torch.manual_seed(c['seed']); np.random.seed(c['seed'])
smp._randn_like = lambda x: torch.randn(x.shape).to(x.device)
out = smp.sample(Wrap(pre), flags.cuda())
for i, (o, g) in enumerate(zip(rec_o, rec_g)):
    print(i, 'sigma %.5g/%.5g' % (o[1], g[1]), 'coin', o[2], 'x_in err %.3g (scale %.3g)' % (float((o[0]-g[0]).abs().max()), float(o[0].abs().max())),
          'D err %.3g (scale %.3g)' % (float((o[3]-g[2]).abs().max()), float(o[3].abs().max())))
print('final err', float((out-ref).abs().max()))
