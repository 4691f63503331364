"""Phase timeline of k_wattn_res from in-kernel s_memtime stamps (diagnostic build only: SWG_LIB must point at a
library built with the STAMP() instrumentation, which exports swg_debug_stamps).  usage: diag_stamps.py [B]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import _lib as L, engine as E
torch.set_grad_enabled(False)
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 112
Cd = int(sys.argv[2]) if len(sys.argv) > 2 else 96
res, heads = {96: (96, 3), 192: (48, 6), 384: (24, 12)}[Cd]
blk = S.SwinTransformerBlock(Cd, 512, (res, res), heads, 6, 0).to(dev)
x = torch.randn(B, res * res, Cd, device=dev); out = torch.empty_like(x)
cond = torch.randn(B, 2 * Cd, device=dev) * 0.1
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
keep = E.Keep(); d = E.block_desc(keep, blk, 0)
for _ in range(5):
    L.check(L.lib.swg_wattn_fwd(C.byref(d), B, x.data_ptr(), cond.data_ptr(), 2 * Cd, out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'attn')
torch.cuda.synchronize()
fn = L.lib.swg_debug_stamps
fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
buf = (C.c_ulonglong * (4 * 64))()
assert fn(buf, 4 * 64) == 0
names = {0: 'top', 1: 'prologue done', 2: 'ring primed', 27: 'heads done', 28: 'stores issued'}
for h in range(3):
    for i, n in enumerate(['head top', 'V done', 'Q done', 'K done', 'S+bias', 'softmax', 'PV', ]):
        names[3 + 8 * h + i] = f'h{h} {n}'
for slot in range(4):
    t = [buf[slot * 64 + i] for i in range(64)]
    print(f'--- slot {slot}: window total {t[27] - t[0]} ticks')
    prev = t[0]
    for i in sorted(names):
        if t[i]:
            print(f'   {names[i]:16s} +{t[i] - prev:7d}   (at {t[i] - t[0]:7d})')
            prev = t[i]
