"""Microbenchmark the two halves of a Swin block (swg_wattn_fwd / swg_ln_mlp_fwd) per stage geometry."""
import ctypes as C, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import _lib as L, engine as E
torch.set_grad_enabled(False)
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
which = sys.argv[2] if len(sys.argv) > 2 else 'both'
stages = [(96, 96, 3), (48, 192, 6), (24, 384, 12), (12, 768, 24)]
ws = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for res, Cd, heads in stages:
    for shift in ([0, 3] if res == 24 else [0]):
        blk = S.SwinTransformerBlock(Cd, 512, (res, res), heads, 6, shift).to(dev)
        T = B * res * res
        x = torch.randn(B, res * res, Cd, device=dev); out = torch.empty_like(x)
        cond = torch.randn(B, 2 * Cd, device=dev) * 0.1
        for fused in ([True, False] if (Cd in E.FUSED_MLP_DIMS or Cd in E.STREAM_MLP_HC) else [False]):
            E.USE_FUSED_MLP = fused
            keep = E.Keep(); d = E.block_desc(keep, blk, 0)
            if which in ('both', 'mlp'):
                us = timeit(lambda: L.check(L.lib.swg_ln_mlp_fwd(C.byref(d), B, x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'mlp'))
                print(f"C={Cd:4d} res={res:3d} T={T:7d} mlp   fused={fused!s:5s} {us:8.1f} us  {16*T*Cd*Cd/us/1e6:7.1f} TF/s  {8*T*Cd/us/1e3:7.1f} GB/s(min)")
        if which in ('both', 'attn'):
            us = timeit(lambda: L.check(L.lib.swg_wattn_fwd(C.byref(d), B, x.data_ptr(), cond.data_ptr(), 2 * Cd, out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'attn'))
            n = 36
            print(f"C={Cd:4d} res={res:3d} T={T:7d} attn  shift={shift}     {us:8.1f} us  {(8*T*Cd*Cd+4*T*n*Cd)/us/1e6:7.1f} TF/s  {8*T*Cd/us/1e3:7.1f} GB/s(min)")
            if len(sys.argv) > 3:
                from swingnn_amd import prof
                with prof.profiled():
                    for _ in range(5): L.check(L.lib.swg_wattn_fwd(C.byref(d), B, x.data_ptr(), cond.data_ptr(), 2 * Cd, out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'attn')
                print("   ", {k: round(v['ms'] / 5 * 1e3, 1) for k, v in prof.read().items()})
