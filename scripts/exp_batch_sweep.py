"""Per-GPU-batch sweep of one network pass as the product path launches it (engine.launch_batch decides the split), and
with the one-pass kernels' token gates rescaled (SWG_FUSED_MIN_TOKENS_PCT, read once per process: run once per value).
usage: python scripts/exp_batch_sweep.py [config] [batches...]   -> one JSON line per batch"""
import json, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from swingnn_amd import engine as E
torch.set_grad_enabled(False)
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'grid'
batches = [int(b) for b in sys.argv[2:]] or [28, 56, 58, 84, 112]
cfg = bench.CONFIGS[cfgname]
dev = torch.device('cuda:0')
model = bench.build_model(cfg, dev)
np.random.rand = lambda: 0.9          # one network pass per call
cache = None
for B in batches:
    flags = bench.make_flags(cfg, B, np.random.default_rng(0)).to(dev)
    x = torch.randn(B, cfg['img_size'], cfg['img_size'], device=dev); x = x + x.transpose(1, 2)
    sig = torch.full((1,), 3.0, device=dev).expand(B)
    for _ in range(3): model(x, flags, sig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): model(x, flags, sig)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    cache = E.get_cache(model.model)
    print(json.dumps(dict(config=cfgname, batch=B, graphs_per_launch=E.launch_batch(B, cfg['img_size'], cache),
                          fused_ops=cache.fused_ops(E.launch_batch(B, cfg['img_size'], cache)),
                          gate_pct=os.environ.get("SWG_FUSED_MIN_TOKENS_PCT", "100"),
                          ms_per_pass=round(ms, 3), us_per_graph=round(ms / B * 1e3, 2))), flush=True)
