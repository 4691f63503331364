"""Phase timeline of k_merge_linear from in-kernel s_memtime stamps (diagnostic build: bash scripts/build_diag.sh
merge_fused, then SWG_LIB=swingnn_amd/libswingnn_hip_diag.so).  usage: diag_merge_stamps.py [B] [C]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import _lib as L, engine as E
torch.set_grad_enabled(False)
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 56
Cd = int(sys.argv[2]) if len(sys.argv) > 2 else 96
res = {96: 96, 192: 48}[Cd]
m = S.PatchMerging((res, res), Cd).to(dev)
x = torch.randn(B, res * res, Cd, device=dev)
out = torch.empty(B, res * res // 4, 2 * Cd, device=dev)
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
keep = E.Keep(); d = E.merge_desc(keep, m)
for _ in range(5):
    L.check(L.lib.swg_patch_merge_fwd(C.byref(d), B, x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()), 'merge')
torch.cuda.synchronize()
fn = L.lib.swg_debug_stamps
fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
buf = (C.c_ulonglong * (4 * 64))()
assert fn(buf, 4 * 64) == 0
names = {0: 'top', 1: 'rows requested', 2: 'g/b in LDS', 3: 'LN0 | primed', 4: 'primed | LN0', 5: 'prologue done', 60: 'ring drained', 61: 'stores issued'}
nch = 2 * Cd // E.MERGE_FUSED_NC[Cd]
for j in range(nch):
    names[6 + 2 * j] = f'chunk {j} landed'
    names[7 + 2 * j] = f'chunk {j} done'
order = sorted(names)
for slot in range(3):
    t = [buf[slot * 64 + i] for i in range(64)]
    print(f'--- slot {slot}: total {t[61] - t[0]} ticks')
    prev = t[0]
    for i in order:
        if t[i]:
            print(f'   {names[i]:18s} +{t[i] - prev:7d}   (at {t[i] - t[0]:7d})')
            prev = t[i]
