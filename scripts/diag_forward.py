"""Diagnostic: per-case forward error of the HIP path vs golden (run on the GPU box)."""
import json, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swingnn_amd as S
from tests import helpers as H
from tests.golden.cases import NET_CASES
from tests.test_gpu_parity import build_precond
torch.set_grad_enabled(False)
g = np.load('tests/golden/golden.npz'); meta = json.load(open('tests/golden/golden_meta.json'))
for name in ['ego60', 'grid192']:
    c = NET_CASES[name]
    for regime in c['regimes']:
        pre = build_precond(S, name, meta, regime)
        for si in range(len(c['sigmas'])):
            flags, x, sig, sc = H.net_inputs(c, si)
            cn = sig.log()/4
            F0 = pre.model(x.cuda(), flags.cuda(), cn.cuda(), None).cpu()
            F1 = pre.model(x.cuda(), flags.cuda(), cn.cuda(), sc.cuda()).cpu()
            r0 = torch.from_numpy(g[f'net/{name}/{regime}/s{si}/F_nosc']); r1 = torch.from_numpy(g[f'net/{name}/{regime}/s{si}/F_sc'])
            print(name, regime, 'sigma', c['sigmas'][si], 'F scale %.3g' % float(r0.abs().max()),
                  'err nosc %.3g' % float((F0-r0).abs().max()), 'err sc %.3g' % float((F1-r1).abs().max()),
                  'mean err %.3g' % float((F0-r0).abs().mean()))
