"""Per-kernel breakdown of one preconditioned network pass (run on the GPU box).
usage: python scripts/prof_forward.py [config] [batch] [reps]"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from swingnn_amd import prof
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'grid'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.set_grad_enabled(False)
cfg = bench.CONFIGS[cfgname]
dev = torch.device('cuda:0')
model = bench.build_model(cfg, dev)
flags = bench.make_flags(cfg, B, np.random.default_rng(0)).to(dev)
x = torch.randn(B, cfg['img_size'], cfg['img_size'], device=dev); x = x + x.transpose(1, 2)
sig = torch.full((1,), 3.0, device=dev).expand(B)
np.random.rand = lambda: 0.9   # single pass per call
for _ in range(3): model(x, flags, sig)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): model(x, flags, sig)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
with prof.profiled():
    for _ in range(reps): model(x, flags, sig)
k = prof.read()
tot = sum(v['ms'] for v in k.values()) / reps
print(f"{cfgname} B={B}: wall {wall*1e3:.3f} ms/pass ({wall/B*1e6:.1f} us/graph), kernel sum {tot:.3f} ms/pass")
for name, v in sorted(k.items(), key=lambda kv: -kv[1]['ms']):
    print(f"  {name:22s} {v['ms']/reps*1e3:9.1f} us/pass  {v['launches']//reps:4d} launches  {v['flops']/(v['ms']*1e-3)/1e12:7.1f} TF/s {v['bytes']/(v['ms']*1e-3)/1e9:8.1f} GB/s")
