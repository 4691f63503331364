"""Phase timeline of k_mlp_stream (PROJ form through swg_block_fwd) from in-kernel s_memtime stamps (diagnostic build:
bash scripts/build_diag.sh mlp_stream; SWG_LIB=swingnn_amd/libswingnn_hip_diag.so).  usage: diag_mlp_stamps.py [B] [C]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import _lib as L, engine as E
torch.set_grad_enabled(False)
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 56
Cd = int(sys.argv[2]) if len(sys.argv) > 2 else 384
res, heads = {192: (48, 6), 384: (24, 12)}[Cd]
blk = S.SwinTransformerBlock(Cd, 512, (res, res), heads, 6, 0).to(dev)
x = torch.randn(B, res * res, Cd, device=dev); out = torch.empty_like(x)
cond = torch.randn(B, 2 * Cd, device=dev) * 0.1
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
keep = E.Keep(); d = E.block_desc(keep, blk, 0)
for _ in range(5):
    L.check(L.lib.swg_block_fwd(C.byref(d), B, x.data_ptr(), cond.data_ptr(), 2 * Cd, out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'block')
torch.cuda.synchronize()
from swingnn_amd import prof
with prof.profiled():
    for _ in range(5):
        L.check(L.lib.swg_block_fwd(C.byref(d), B, x.data_ptr(), cond.data_ptr(), 2 * Cd, out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'block')
    torch.cuda.synchronize()
for k, v in prof.read().items():
    print(f"{k}: {v['ms'] / v['launches'] * 1e3:.1f} us per launch")
fn = L.lib.swg_debug_stamps
fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
buf = (C.c_ulonglong * (4 * 64))()
assert fn(buf, 4 * 64) == 0
names = {0: 'top', 1: 'shortcut in acc', 2: 'proj chunks done', 3: 'LN2 done', 60: 'chunk loop done', 61: 'stores issued'}
for j in range(50):
    names[4 + j] = f'chunk {j} top'
for slot in range(4):
    t = [buf[slot * 64 + i] for i in range(64)]
    print(f'--- slot {slot}: total {t[61] - t[0]} ticks (s_memtime: 100 MHz counter ticks x ?)')
    prev = t[0]
    for i in sorted(names):
        if t[i]:
            print(f'   {names[i]:18s} +{t[i] - prev:7d}   (at {t[i] - t[0]:7d})')
            prev = t[i]
