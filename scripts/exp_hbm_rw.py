"""HBM rate of plain fills / copies at the sizes of the network's token streams (run on the GPU box)."""
import torch
dev = 'cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (50, 198, 396, 1024):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.randn(n, device=dev)
    us = t(lambda: a.fill_(1.5)); print(f"fill  {mb:5d} MB {us:8.1f} us {mb/us*1e3:8.1f} GB/s written")
    us = t(lambda: a.copy_(b)); print(f"copy  {mb:5d} MB {us:8.1f} us {2*mb/us*1e3:8.1f} GB/s read+written")
    us = t(lambda: b.sum()); print(f"sum   {mb:5d} MB {us:8.1f} us {mb/us*1e3:8.1f} GB/s read")
