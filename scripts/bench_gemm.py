"""Microbenchmark swg_linear_bf16 per shape (run on the GPU box)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from swingnn_amd import _lib as L, engine as E
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = []
for (L0, C) in [(576, 384), (144, 768)]:
    M = B * L0
    shapes += [(M, 3*C, C, 'qkv bf16'), (M, C, C, 'proj f32+res'), (M, 4*C, C, 'fc1 gelu bf16'), (M, C, 4*C, 'fc2 f32+res')]
# merge (4C -> 2C at res/2) and breakup (pre: D -> D at res; post: D/4 -> D/4 at 2 res) linears of the grid net
for (L0, C) in [(2304, 96), (576, 192), (144, 384)]:
    shapes += [(B * L0, 2*C, 4*C, 'merge f32')]
for (L0, D) in [(144, 1536), (576, 768), (2304, 384)]:
    shapes += [(B * L0, D, D, 'pre f32'), (4 * B * L0, D // 4, D // 4, 'post f32')]
shapes += [(8192, 8192, 4096, 'big bf16'), (8192, 8192, 4096, 'big f32')]
for M, N, K, kind in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    f32 = 'f32' in kind
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    res = torch.randn(M, N, device=dev) if 'res' in kind else None
    def run():
        L.check(L.lib.swg_linear_bf16(M, N, K, a.data_ptr(), K, w.data_ptr(), bias.data_ptr(), 1 if 'gelu' in kind else 0,
                E._ptr(res), out.data_ptr() if f32 else None, None if f32 else out.data_ptr(), N, E.stream_ptr()), 'lin')
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    by = 2*M*K + 2*N*K + (4 if f32 else 2)*M*N + (4*M*N if res is not None else 0)
    ref = (a.float() @ w.float().t() + bias)
    if 'gelu' in kind: ref = torch.nn.functional.gelu(ref)
    if res is not None: ref = ref + res
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    print(f"M={M:7d} N={N:5d} K={K:5d} {kind:14s} {us:9.1f} us  {2*M*N*K/us/1e6:8.1f} TF/s  {by/us/1e3:8.1f} GB/s  relerr {err:.2e}")
