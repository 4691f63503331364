"""Microbenchmark swg_linear_wgrad_bf16 (dW[N,K] += dY[rows,N]^T X[rows,K]) on the weight-gradient shapes of the grid
dim-96 train step (run on the GPU box).  usage: bench_wgrad.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swingnn_amd import _lib as L, engine as E  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 80
shapes = []
for L0, C in [(9216, 96), (2304, 192), (576, 384), (144, 768)]:
    rows = B * L0
    shapes += [(rows, 3 * C, C, "qkv"), (rows, C, C, "proj"), (rows, 4 * C, C, "fc1"), (rows, C, 4 * C, "fc2")]
for L0, C in [(2304, 96), (576, 192), (144, 384)]:
    shapes += [(B * L0, 2 * C, 4 * C, "merge")]
for rows, N, K, kind in shapes:
    dy = torch.randn(rows, N, device=dev).to(torch.bfloat16)
    x = torch.randn(rows, K, device=dev).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    wsb = int(L.lib.swg_linear_wgrad_workspace_bytes(N, K))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def run():
        L.check(L.lib.swg_linear_wgrad_bf16(rows, N, K, dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), ws.data_ptr(), wsb,
                                            E.stream_ptr()), "wgrad")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    by = 2 * rows * (N + K) + 4 * N * K
    dw.zero_()
    run()
    ref = dy.float().t() @ x.float()
    err = float((dw - ref).abs().max() / ref.abs().max())
    print(f"rows={rows:7d} N={N:5d} K={K:5d} {kind:6s} {us:8.1f} us  {2 * rows * N * K / us / 1e6:7.1f} TF/s  "
          f"{by / us / 1e3:7.1f} GB/s  relerr {err:.1e}")
