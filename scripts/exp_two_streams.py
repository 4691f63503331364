"""Experiment: one network pass at batch B on one stream vs two half-batches on two streams (tail overlap)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 112
torch.set_grad_enabled(False)
cfg = bench.CONFIGS['grid']
dev = torch.device('cuda:0')
model = bench.build_model(cfg, dev)
flags = bench.make_flags(cfg, B, np.random.default_rng(0)).to(dev)
x = torch.randn(B, cfg['img_size'], cfg['img_size'], device=dev); x = x + x.transpose(1, 2)
sig = torch.full((1,), 3.0, device=dev)
np.random.rand = lambda: 0.9
def one(n):
    for _ in range(n): model(x, flags, sig.expand(B))
def split(n, parts, streams):
    cur = torch.cuda.current_stream()
    h = B // parts
    for _ in range(n):
        ev = torch.cuda.Event(); ev.record(cur)
        for i, s in enumerate(streams):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                model(x[i*h:(i+1)*h], flags[i*h:(i+1)*h], sig.expand(h))
            e2 = torch.cuda.Event(); e2.record(s); cur.wait_event(e2)
def timeit(fn, n=10):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"B={B}: single stream {timeit(one):.3f} ms/pass")
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    print(f"   {parts} streams x B={B//parts}: {timeit(lambda n: split(n, parts, streams)):.3f} ms/pass")
