"""Time PatchBreakup (with the concat) per shape, one-pass kernel vs concat + GEMM + tail (run on the GPU box).
usage: python3 scripts/bench_expand.py [batch]"""
import ctypes as C
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import _lib as L, engine as E
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 56
for (dim, res) in [(384, 48), (768, 24), (1536, 12)]:
    u = S.PatchBreakup((res, res), dim).to(dev)
    x = torch.randn(B, res * res, dim // 2, device=dev); sk = torch.randn_like(x)
    out = torch.empty(B, 4 * res * res, dim // 4, device=dev)
    ws = torch.empty(64 * x.numel() + (1 << 20), dtype=torch.uint8, device=dev)
    for fused in (True, False):
        E.USE_FUSED_MLP = fused
        keep = E.Keep(); desc = E.breakup_desc(keep, u)
        def run():
            L.check(L.lib.swg_patch_expand_fwd(C.byref(desc), B, x.data_ptr(), sk.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                               ws.numel(), E.stream_ptr()), 'expand')
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        by = 2 * x.numel() * 4 + out.numel() * 4
        print(f"Cin={dim:5d} res={res:3d} B={B} {'one pass' if fused and desc.img_pre else 'unfused':10s} {us:8.1f} us  {by/us/1e3:7.1f} GB/s")
    E.USE_FUSED_MLP = True
