"""Experiment: ms per network pass and per graph for (batch, parts) combinations of engine.network_pass."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from swingnn_amd import engine as E
torch.set_grad_enabled(False)
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'grid'
cfg = bench.CONFIGS[cfgname]
dev = torch.device('cuda:0')
model = bench.build_model(cfg, dev)
np.random.rand = lambda: 0.9
combos = {'grid': [(112, 1), (112, 2), (168, 2), (168, 3), (224, 2), (224, 4), (280, 5), (84, 3), (56, 2), (56, 1)],
          'dd': [(56, 1), (56, 2), (64, 1), (64, 2), (128, 2), (112, 2), (32, 1)]}[cfgname]
for B, parts in combos:
    E.SPLIT_STREAMS = parts
    E.SPLIT_MIN_WORK = 0
    flags = bench.make_flags(cfg, B, np.random.default_rng(0)).to(dev)
    x = torch.randn(B, cfg['img_size'], cfg['img_size'], device=dev); x = x + x.transpose(1, 2)
    sig = torch.full((1,), 3.0, device=dev).expand(B)
    for _ in range(3): model(x, flags, sig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): model(x, flags, sig)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"B={B:4d} parts={parts}: {ms:7.3f} ms/pass  {ms / B * 1e3:6.1f} us/graph", flush=True)
