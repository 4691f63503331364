"""Host-side plumbing between the nn.Module parameter tree and libswingnn_hip.so.

A ``NetCache`` holds everything the kernels consume that is DERIVED from the fp32 nn.Parameters:
bf16 / K-padded copies of the Linear weights, the stacked affine layers, the collapsed read-out and
the C descriptor structs (include/swingnn_hip.h).  It is rebuilt whenever a parameter's version
counter, storage or device changes.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

import torch

from . import _lib as L


def eff_window(res: int, window: int, shift: int):
    """Window clamp rule of the reference block constructor (model/swin_gnn/swin_gnn.py:189-192)."""
    if res <= window:
        return res, 0
    return window, shift


def _kp(k: int) -> int:
    return (k + 31) // 32 * 32


def _ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_WS = {}
_WS_PRIVATE = None   # while graphed_pass records: {(device, stream): buffer owned by the graph entry}


def workspace(device, nbytes: int) -> torch.Tensor:
    """Scratch buffer from torch's caching allocator, one per (device, stream): passes issued on different streams
    (a sampler on a side stream next to training, the two half-batch streams of network_pass) never share scratch.  A
    buffer that is outgrown is handed back to the allocator only after the work queued on its stream (record_stream).
    While a HIP graph is being recorded the buffers come from the recording's own set instead (graphed_pass): a graph
    bakes the scratch ADDRESS into its nodes, so that memory must live as long as the graph, not as long as the pool
    entry."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx)
    key = (idx, stream.cuda_stream)
    if _WS_PRIVATE is not None:
        buf = _WS_PRIVATE.get(key)
        if buf is None or buf.numel() < nbytes:
            raise RuntimeError("graph recording: no private workspace of the needed size for this stream")
        return buf
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            buf.record_stream(stream)
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _f32(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


class Keep:
    """Collects the tensors / ctypes objects a descriptor points into."""

    def __init__(self, registry=None):
        self.items = []
        self.rpb_tiles = []   # (rpb fp32 view, heads, window, tiles) of blocks with swg_block.rpb_tiles
        # when not None: every derived bf16 operand is recorded as (mode, sources, dst[, idx]) so that it can be
        # rewritten in place after an optimizer step (RefreshTable / swg_refresh_derived)
        self.registry = registry

    def __call__(self, obj):
        self.items.append(obj)
        return obj

    def image(self, fn, srcs, *args):
        """A fragment-order image fn(*srcs, *args) of one or two weights.  Registered as a gather: the index map is
        fn evaluated on element numbers (the builders only permute / gather / zero-pad, exact in fp32 below 2^24)."""
        img = self(fn(*srcs, *args))
        if self.registry is not None:
            off, fake = 1, []
            for w in srcs:
                n = w.numel()
                assert off + n < (1 << 24)
                fake.append(torch.arange(off, off + n, dtype=torch.float32, device=w.device).view(w.shape))
                off += n
            idx = self(fn(*fake, *args, dtype=torch.float32).round().to(torch.int32).contiguous())
            assert idx.shape == img.shape
            self.registry.append(("gather", tuple(srcs), img, idx))
        return img

    def image_pair(self, fn_a, srcs_a, args_a, fn_b, srcs_b, args_b):
        """Two images in ONE allocation, b directly behind a (a kernel streams across the seam); each half is
        registered for the in-place refresh on its own.  -> (whole, view of a, view of b)"""
        a, b = fn_a(*srcs_a, *args_a), fn_b(*srcs_b, *args_b)
        whole = self(torch.cat([a.reshape(-1), b.reshape(-1)]))
        va, vb = whole[:a.numel()], whole[a.numel():]
        if self.registry is not None:
            for fn, srcs, args, view in ((fn_a, srcs_a, args_a, va), (fn_b, srcs_b, args_b, vb)):
                off, fake = 1, []
                for w in srcs:
                    n = w.numel()
                    assert off + n < (1 << 24)
                    fake.append(torch.arange(off, off + n, dtype=torch.float32, device=w.device).view(w.shape))
                    off += n
                idx = self(fn(*fake, *args, dtype=torch.float32).round().to(torch.int32).reshape(-1).contiguous())
                assert idx.numel() == view.numel()
                self.registry.append(("gather", tuple(srcs), view, idx))
        return whole, va, vb

    def f32(self, t):
        return _ptr(self(_f32(t)))

    def lin(self, weight, bias=None) -> L.SwgLin:
        w = weight.detach()
        N, K = w.shape
        Kp = _kp(K)
        wb = torch.zeros((N, Kp), dtype=torch.bfloat16, device=w.device)
        wb[:, :K] = w
        self(wb)
        if self.registry is not None:
            self.registry.append(("pad", (weight,), wb, None))
        return L.SwgLin(_ptr(wb), None if bias is None else self.f32(bias), N, K, Kp)

    def ln(self, mod) -> L.SwgLn:
        return L.SwgLn(self.f32(mod.weight), self.f32(mod.bias))


class RefreshTable:
    """Device tables for swg_refresh_derived built from a Keep registry: one launch rewrites every registered bf16
    operand from the current fp32 parameters."""

    CHUNK = 16384

    def __init__(self, registry, device):
        items = (L.SwgRefreshItem * max(len(registry), 1))()
        ci, c0, c1 = [], [], []
        self.keep = []
        for k, (mode, srcs, dst, idx) in enumerate(registry):
            w = srcs[0]
            if any((not t.is_contiguous()) or t.dtype != torch.float32 for t in srcs):
                raise NotImplementedError("derived operands are refreshed from contiguous fp32 parameters")
            if mode == "gather":
                n0 = w.numel()
                items[k] = L.SwgRefreshItem(w.data_ptr(), srcs[1].data_ptr() if len(srcs) > 1 else None, dst.data_ptr(),
                                            idx.data_ptr(), 0, n0, 0, 0, 0, 0, dst.numel())
                n = (dst.numel() + self.CHUNK - 1) // self.CHUNK
                ci += [k] * n; c0 += list(range(n)); c1 += [0] * n
                self.keep.append(idx)
            elif mode == "pad":      # dst [N, ld] <- w viewed as [N, K]
                N, ld = dst.shape
                K = w.numel() // N
                items[k] = L.SwgRefreshItem(w.data_ptr(), None, dst.data_ptr(), None, 1, 0, N, K, ld, 0, dst.numel())
                n = (dst.numel() + self.CHUNK - 1) // self.CHUNK
                ci += [k] * n; c0 += list(range(n)); c1 += [0] * n
            else:                    # "padT": dst [K, ld] <- w [N, K] transposed
                N, K = w.shape
                ld = dst.shape[1]
                items[k] = L.SwgRefreshItem(w.data_ptr(), None, dst.data_ptr(), None, 2, 0, N, K, ld, 0, dst.numel())
                tk, tn = (K + 63) // 64, (ld + 63) // 64
                for a in range(tk):
                    ci += [k] * tn; c0 += [a] * tn; c1 += list(range(tn))
            self.keep += list(srcs) + [dst]
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.items = raw.to(device)
        i32 = dict(dtype=torch.int32, device=device)
        self.ci, self.c0, self.c1 = torch.tensor(ci, **i32), torch.tensor(c0, **i32), torch.tensor(c1, **i32)
        self.n_chunks = len(ci)

    def run(self):
        if self.n_chunks:
            L.check(L.lib.swg_refresh_derived(self.items.data_ptr(), self.ci.data_ptr(), self.c0.data_ptr(),
                                              self.c1.data_ptr(), self.n_chunks, stream_ptr()), "swg_refresh_derived")


FUSED_MLP_DIMS = (96, 60, 120) if os.environ.get("SWG_MLP_FUSED_PAD", "1") == "1" else (96,)   # 60 / 120: A/B knob
USE_FUSED_MLP = True


def _slot_swizzle(t: torch.Tensor) -> torch.Tensor:
    """t [..., rows, 4, 8]: move 16-byte chunk ch of row r to slot ch ^ ((-(r >> 2)) & 3) (the kernels'
    slot_swz): conflict-free ds_read_b128 fragment reads from 64-byte LDS rows."""
    rows = t.shape[-3]
    r = torch.arange(rows, device=t.device)
    f = (-(r >> 2)) & 3
    idx = torch.arange(4, device=t.device)[None, :] ^ f[:, None]          # out[r, slot] = in[r, slot ^ f(r)]
    return torch.gather(t, -2, idx[:, :, None].expand(*t.shape[:-3], rows, 4, 8))


def mlp_image(w1: torch.Tensor, w2: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """Weight image of the fused MLP kernel (csrc/mlp_fused.hip): per 64-wide hidden chunk, the fc1 rows
    [64, C] as [k-step][row][4 slots][8] followed by the fc2 columns [C, 64] as [k-step][row][4 slots][8] with
    the k order the fc1 accumulators imply (and the fc2 rows permuted, see below): element e of chunk ch is hidden 32*s + (4ch+e if e<4 else 16+4ch+e-4)."""
    w1, w2 = w1.detach().float(), w2.detach().float()
    H, Cdim = w1.shape
    if Cdim % 32:   # C = 60 / 120: zero-padded to 64 / 128 channel slots and 4 x that many hidden features
        cs = -(-Cdim // 32) * 32
        p1 = torch.zeros(4 * cs, cs, device=w1.device)
        p1[:H, :Cdim] = w1
        p2 = torch.zeros(cs, 4 * cs, device=w1.device)
        p2[:Cdim, :H] = w2
        w1, w2, H, Cdim = p1, p2, 4 * cs, cs
    nch, ks = H // 64, Cdim // 32
    a = w1.reshape(nch, 64, ks, 4, 8).permute(0, 2, 1, 3, 4)              # [hc, s, r, ch, e]
    a = _slot_swizzle(a.contiguous()).reshape(nch, -1)
    ch = torch.arange(4, device=w1.device)[:, None]
    e = torch.arange(8, device=w1.device)[None, :]
    pi = torch.where(e < 4, 4 * ch + e, 16 + 4 * ch + e - 4).reshape(-1)  # [ch*8+e] -> k within a 32-step
    # image row (nt*16 + i) of the fc2 part is channel 32*(nt>>1) + 8*(i>>2) + 4*(nt&1) + (i&3): the accumulator
    # rows of a lane then coincide with the channels it loaded for fc1 (residual add from registers)
    img_row = torch.arange(Cdim, device=w1.device)
    nt, i = img_row // 16, img_row % 16
    rho = 32 * (nt // 2) + 8 * (i // 4) + 4 * (nt % 2) + (i % 4)
    b = w2[rho].reshape(Cdim, nch, 2, 32)[..., pi].reshape(Cdim, nch, 2, 4, 8).permute(1, 2, 0, 3, 4)  # [hc, s2, row, ch, e]
    b = _slot_swizzle(b.contiguous()).reshape(nch, -1)
    return torch.cat([a, b], dim=1).to(dtype).contiguous()


MLP_BWD_FRONT_HC = {96: 64, 192: 32, 384: 32}   # hidden chunk width of csrc/mlp_bwd_front.hip per channel count


def mlp_bwd_front_image(w1: torch.Tensor, w2: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """Weight image of csrc/mlp_bwd_front.hip: per hidden chunk of hcw features the fc1 rows [hcw, C] and the fc2^T
    rows [hcw, C], each as [k-step][row][4 slots][8] in the swizzled LDS row layout of mlp_image."""
    w1, w2t = w1.detach().float(), w2.detach().float().t().contiguous()
    H, Cdim = w1.shape
    hcw = MLP_BWD_FRONT_HC[Cdim]
    nch, ks = H // hcw, Cdim // 32
    parts = []
    for w in (w1, w2t):
        a = w.reshape(nch, hcw, ks, 4, 8).permute(0, 2, 1, 3, 4)              # [hc, s, r, ch, e]
        parts.append(_slot_swizzle(a.contiguous()).reshape(nch, -1))
    return torch.cat(parts, dim=1).to(dtype).contiguous()


MLP_BWD_FRONT_DIMS = (96, 192, 384)   # channel counts with the fused front half of the MLP backward
STREAM_MLP_HC = {384: 32, 192: 32}  # hidden chunk width of csrc/mlp_stream.hip per channel count


def mlp_stream_image(w1: torch.Tensor, w2: torch.Tensor, hc: int, dtype=torch.bfloat16) -> torch.Tensor:
    """Weight image of csrc/mlp_stream.hip, in MFMA fragment order (1 KiB = 64 lanes x 8 bf16 per fragment, lane
    16*lg + i holds row i, k-slots 8*lg .. 8*lg+7).  Image chunk j (j = 0 .. 4C/hc + 1) is
      [fc1 rows of hidden chunk j   as fragments (k-step, hidden tile)]
      [fc2 cols of hidden chunk j-2 as fragments (k-step, rho-ordered out tile), k-slot order pi]
    with zero blocks where the hidden chunk does not exist (the kernel's loop is a 3-stage pipeline)."""
    w1, w2 = w1.detach().float(), w2.detach().float()
    H, Cdim = w1.shape
    nch, ks, ht, s2, nt = H // hc, Cdim // 32, hc // 16, hc // 32, Cdim // 16
    dev = w1.device
    # fc1 part: [chunk, ks, ht, lg, i, e] <- w1[chunk*hc + ht*16 + i, ks*32 + lg*8 + e]
    a = w1.reshape(nch, ht, 16, ks, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(nch, -1)
    # fc2 part: [chunk, s2, nt, lg, i, e] <- w2[rho(nt*16 + i), chunk*hc + s2*32 + pi(lg, e)]
    b = w2[_rho(Cdim, dev)].reshape(nt, 16, nch, s2, 32)[..., _pi(dev)].reshape(nt, 16, nch, s2, 4, 8)
    b = b.permute(2, 3, 0, 4, 1, 5).reshape(nch, -1)
    z = torch.zeros(2, a.shape[1], device=dev)
    img = torch.cat([torch.cat([a, z], 0), torch.cat([z, b], 0)], dim=1)  # [nch + 2, W1 part | W2 part]
    return img.to(dtype).contiguous()


def proj_frag_image(proj_w: torch.Tensor, heads: int, dtype=torch.bfloat16) -> torch.Tensor:
    """The attention projection weight [C, C] as MFMA fragments for the PROJ form of csrc/mlp_stream.hip: fragment
    (k-step ks, out tile nt), k-step major, holds rows rho(nt*16 + i) and input columns ks*32 + lg*8 + e in the slot
    order in which k_wattn_ring stores o (slot h*32 + 8lg + e = head dim pi(lg, e))."""
    proj_w = proj_w.detach().float()
    Cdim = proj_w.shape[0]
    dev = proj_w.device
    hd = Cdim // heads
    if hd != 32:   # head_dim 20 (k_mlp_fused PROJ form): rows padded to 32 | Cs, every head's columns to a 32-slot group
        cs = -(-Cdim // 32) * 32
        pw = torch.zeros(cs, heads, 32, device=dev)
        pw[:Cdim, :, :hd] = proj_w.reshape(Cdim, heads, hd)
        wp = pw[..., _pi(dev)].reshape(cs, heads * 32)
        w = wp[_rho(cs, dev)].reshape(cs // 16, 16, heads, 4, 8)                    # [nt, i, ks, lg, e]
        return w.permute(2, 0, 3, 1, 4).reshape(-1).to(dtype).contiguous()
    wp = proj_w.reshape(Cdim, heads, 32)[..., _pi(dev)].reshape(Cdim, Cdim)        # columns in o's slot order
    w = wp[_rho(Cdim, dev)].reshape(Cdim // 16, 16, Cdim // 32, 4, 8)               # [nt, i, ks, lg, e]
    return w.permute(2, 0, 3, 1, 4).reshape(-1).to(dtype).contiguous()             # [ks, nt, lg, i, e]


# csrc/wattn_ring.hip at C = 192 with the projection fused (round-2 form) instead of folded into the MLP kernel: A/B knob,
# read by the library too (wattn_ring_fuses_proj)
RING192_FUSE_PROJ = os.environ.get("SWG_RING192_FUSE_PROJ", "0") == "1"
# channel counts with the fused attention-tail + MLP kernel (csrc/mlp_stream.hip, PROJ form)
PROJ_MLP_DIMS = (384,) if RING192_FUSE_PROJ else (384, 192)
USE_FUSED_ATTN = True
MLP_FUSED_PROJ = os.environ.get("SWG_MLP_FUSED_PROJ", "1") == "1"   # A/B knob: attention tail folded into k_mlp_fused (C = 60 / 120)
RING_HD20 = os.environ.get("SWG_RING_HD20", "1") == "1"   # A/B knob: fused attention core for head_dim 20
RING_SMALL_WINDOWS = os.environ.get("SWG_RING_SMALL_WINDOWS", "1") == "1"   # A/B knob: 4x4 / 2x2 windows at C = 192 / 384


def _rho(Cdim: int, device) -> torch.Tensor:
    """Channel held by image row nt*16 + i of an output-feature operand: 32*(nt>>1) + 8*(i>>2) + 4*(nt&1) + (i&3),
    i.e. accumulator row 4*lg + r of tile nt = 2*ks + hh is channel 32*ks + 8*lg + 4*hh + r -- the channels a lane
    loaded for k-step ks, so residual adds and stores need no cross-lane movement."""
    row = torch.arange(Cdim, device=device)
    nt, i = row // 16, row % 16
    return 32 * (nt // 2) + 8 * (i // 4) + 4 * (nt % 2) + (i % 4)


def _pi(device) -> torch.Tensor:
    """k-slot order implied by two stacked 16x16 accumulator tiles: element e of chunk ch is index
    (4ch + e if e < 4 else 16 + 4ch + e - 4) of a 32-deep k-step."""
    ch = torch.arange(4, device=device)[:, None]
    e = torch.arange(8, device=device)[None, :]
    return torch.where(e < 4, 4 * ch + e, 16 + 4 * ch + e - 4).reshape(-1)


RING_ATTN_FUSED_PROJ = {96: True, 192: RING192_FUSE_PROJ, 384: False}  # 96: wattn_res.hip (weights resident in LDS)  # csrc/wattn_ring.hip geometries


def wattn_ring_image(qkv_w: torch.Tensor, proj_w: torch.Tensor, heads: int, fuse_proj: bool,
                     dtype=torch.bfloat16) -> torch.Tensor:
    """Weight stream of csrc/wattn_ring.hip (head_dim 32) as 1-KiB MFMA fragments (lane 16*lg + i holds row i,
    k-slots 8lg..8lg+7): per head the slabs Wv, Wq, Wk (rows [32, C]) as [k-step][16-row half][lane][8] and, when the
    projection is fused, the Wproj column slice [C, 32] as [out tile][lane][8] with rows in rho order and the
    head-dim slots in pi order.  Otherwise the projection weight [C, C] follows the stream, its input columns in
    the order the kernel stores o (slot h*32 + 8lg + e holds head dim pi(lg, e))."""
    qkv_w, proj_w = qkv_w.detach().float(), proj_w.detach().float()
    Cdim = proj_w.shape[0]
    dev = qkv_w.device
    hd = Cdim // heads
    if hd != 32:
        # head_dim 20: every head padded to a 32-slot group (slots hd..31 zero), input channels to the 32-deep k-step
        assert not fuse_proj and hd < 32
        ck = -(-Cdim // 32) * 32
        q = torch.zeros(3, heads, 32, ck, device=dev)
        q[:, :, :hd, :Cdim] = qkv_w.reshape(3, heads, hd, Cdim)
        qkv_w = q.reshape(3 * heads * 32, ck)
        pw = torch.zeros(Cdim, heads, 32, device=dev)
        pw[:, :, :hd] = proj_w.reshape(Cdim, heads, hd)
        proj_w = pw.reshape(Cdim, heads * 32)
    co = heads * 32                       # q / k / v slot rows
    ks = qkv_w.shape[1] // 32
    slabs = []
    for part in (2, 0, 1):  # V, Q, K
        w = qkv_w[part * co:(part + 1) * co].reshape(heads, 2, 16, ks, 4, 8)       # [h, dt, i, ks, lg, e]
        slabs.append(w.permute(0, 3, 1, 4, 2, 5).reshape(heads, -1))               # [h, (ks, dt, lg, i, e)]
    wp = proj_w.reshape(Cdim, heads, 32)[..., _pi(dev)]                             # [c, h, slot]
    if fuse_proj:
        w = wp[_rho(Cdim, dev)].reshape(Cdim // 16, 16, heads, 4, 8)                # [nt, i, h, lg, e]
        slabs.append(w.permute(2, 0, 3, 1, 4).reshape(heads, -1))                   # [h, (nt, lg, i, e)]
        return torch.stack(slabs, dim=1).reshape(-1).to(dtype).contiguous()
    img = torch.stack(slabs, dim=1).reshape(-1)
    return torch.cat([img, wp.reshape(-1)]).to(dtype).contiguous()


def block_desc(keep: Keep, blk, cond_off: int) -> L.SwgBlock:
    img = None
    aimg = None
    # fused attention geometries (csrc/wattn_res.hip: C = 96, windows 4 / 6 / 8; csrc/wattn_ring.hip: C = 192 / 384,
    # windows 6 / 8 and, as the attention core alone, 4 / 2); the weight image does not depend on the window size
    wins = (4, 6, 8) if blk.dim == 96 else ((2, 4, 6, 8) if RING_SMALL_WINDOWS and not RING_ATTN_FUSED_PROJ.get(blk.dim)
                                            else (6, 8))
    if (USE_FUSED_ATTN and blk.dim in RING_ATTN_FUSED_PROJ and blk.window_size in wins
            and blk.dim // blk.num_heads == 32 and blk.input_resolution[0] % blk.window_size == 0):
        aimg = _ptr(keep.image(wattn_ring_image, (blk.attn.qkv.weight, blk.attn.proj.weight), blk.num_heads,
                               RING_ATTN_FUSED_PROJ[blk.dim]))
    elif (USE_FUSED_ATTN and RING_HD20 and blk.dim in (60, 120, 240) and blk.num_heads * 20 == blk.dim
            and blk.window_size in (2, 4, 8) and blk.input_resolution[0] % blk.window_size == 0):
        # head_dim 20 (the dim-60 networks): attention core with 32-slot head groups (csrc/wattn_ring.hip HD = 20)
        aimg = _ptr(keep.image(wattn_ring_image, (blk.attn.qkv.weight, blk.attn.proj.weight), blk.num_heads, False))
    pimg = None
    if USE_FUSED_MLP and blk.dim in FUSED_MLP_DIMS:
        img = _ptr(keep.image(mlp_image, (blk.mlp.fc1.weight, blk.mlp.fc2.weight)))
        if aimg is not None and blk.dim in (60, 120) and MLP_FUSED_PROJ:
            # head_dim-20 blocks: the projection as fragments for the PROJ form of k_mlp_fused (swg_block_fwd)
            pimg = _ptr(keep.image(proj_frag_image, (blk.attn.proj.weight,), blk.num_heads))
    elif (USE_FUSED_MLP and blk.dim in PROJ_MLP_DIMS and aimg is not None and not RING_ATTN_FUSED_PROJ[blk.dim]
          and blk.dim in STREAM_MLP_HC):
        # projection fragments + MLP stream in one allocation: the fused kernel streams across the seam, the plain
        # MLP kernel starts at the second half
        whole, _, vb = keep.image_pair(proj_frag_image, (blk.attn.proj.weight,), (blk.num_heads,),
                                       mlp_stream_image, (blk.mlp.fc1.weight, blk.mlp.fc2.weight),
                                       (STREAM_MLP_HC[blk.dim],))
        pimg, img = _ptr(whole), _ptr(vb)
    elif USE_FUSED_MLP and blk.dim in STREAM_MLP_HC:
        img = _ptr(keep.image(mlp_stream_image, (blk.mlp.fc1.weight, blk.mlp.fc2.weight), STREAM_MLP_HC[blk.dim]))
    rpb = keep(_f32(blk.attn.relative_position_bias_table))
    tiles = None
    if aimg is not None and blk.dim in (192, 384) and blk.window_size == 6 and USE_RPB_TILES and rpb.is_cuda:
        # bias of every head in accumulator layout (swg_rpb_tiles): k_wattn_ring loads it straight into the S registers
        ntm = (blk.window_size ** 2 + 15) // 16
        tiles = keep(torch.empty(blk.num_heads * ntm * ntm * 64 * 4, dtype=torch.float32, device=rpb.device))
        keep.rpb_tiles.append((rpb, blk.num_heads, blk.window_size, tiles))
        L.check(L.lib.swg_rpb_tiles(rpb.data_ptr(), blk.num_heads, blk.window_size, tiles.data_ptr(), stream_ptr()),
                "swg_rpb_tiles")
    return L.SwgBlock(blk.dim, blk.input_resolution[0], blk.window_size, blk.shift_size, blk.num_heads, cond_off,
                      keep.ln(blk.norm1), keep.ln(blk.norm2),
                      keep.lin(blk.attn.qkv.weight, blk.attn.qkv.bias), keep.lin(blk.attn.proj.weight, blk.attn.proj.bias),
                      keep.lin(blk.mlp.fc1.weight, blk.mlp.fc1.bias), keep.lin(blk.mlp.fc2.weight, blk.mlp.fc2.bias),
                      _ptr(rpb), img, aimg, pimg, _ptr(tiles))


USE_RPB_TILES = os.environ.get("SWG_RPB_TILES", "1") == "1"   # A/B knob


def build_rpb_tiles(keep):
    """(Re)build the bias tiles of every block that has them from the current relative_position_bias_table."""
    for rpb, heads, win, tiles in keep.rpb_tiles:
        L.check(L.lib.swg_rpb_tiles(rpb.data_ptr(), heads, win, tiles.data_ptr(), stream_ptr()), "swg_rpb_tiles")


MERGE_FUSED_NC = {96: 32, 192: 16}   # output features per 24-KiB weight chunk of csrc/merge_fused.hip, per input width


def merge_image(w: torch.Tensor, nc: int, dtype=torch.bfloat16) -> torch.Tensor:
    """Weight image of csrc/merge_fused.hip: the reduction weight [2C, 4C] in MFMA fragment order (1 KiB per fragment,
    lane 16*lg + i holds row i, k-slots 8*lg .. 8*lg+7), chunk c = output features [c*nc, (c+1)*nc) as fragments
    (k-step, 16-row output tile), k-step major."""
    w = w.detach().float()
    N, K = w.shape
    a = w.reshape(N // nc, nc // 16, 16, K // 32, 4, 8).permute(0, 3, 1, 4, 2, 5)   # [chunk, ks, ht, lg, i, e]
    return a.reshape(-1).to(dtype).contiguous()


def merge_desc(keep: Keep, m) -> L.SwgMerge:
    img = None
    if USE_FUSED_MLP and m.dim in MERGE_FUSED_NC and m.input_resolution[0] % 2 == 0:
        img = _ptr(keep.image(merge_image, (m.reduction.weight,), MERGE_FUSED_NC[m.dim]))
    return L.SwgMerge(m.dim, m.input_resolution[0], keep.ln(m.norm), keep.lin(m.reduction.weight), img)


EXPAND_FUSED_NC = {384: 32, 768: 16}   # output features per 24-KiB pre_linear chunk of csrc/expand_fused.hip, per 4 C'
EXPAND_FUSED_DIMS = tuple(EXPAND_FUSED_NC)


def expand_post_image(w: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """post_linear weight [C', C'] as MFMA fragments for csrc/expand_fused.hip: fragment (k-step s, out tile nt) holds
    rows nt*16 + i and input columns 32 s + pi(lg, e): the slot order of a pair of stacked accumulator tiles, which the
    kernel feeds back as the operand."""
    w = w.detach().float()
    Co = w.shape[0]
    wp = w.reshape(Co, Co // 32, 32)[..., _pi(w.device)]                              # [row, s, (lg, e)]
    a = wp.reshape(Co // 16, 16, Co // 32, 4, 8).permute(2, 0, 3, 1, 4)              # [s, nt, lg, i, e]
    return a.reshape(-1).to(dtype).contiguous()


def breakup_desc(keep: Keep, u) -> L.SwgBreakup:
    ipre = ipost = None
    if USE_FUSED_MLP and u.dim in EXPAND_FUSED_DIMS:
        ipre = _ptr(keep.image(merge_image, (u.pre_linear.weight,), EXPAND_FUSED_NC[u.dim]))
        ipost = _ptr(keep.image(expand_post_image, (u.post_linear.weight,)))
    return L.SwgBreakup(u.dim, u.input_resolution[0], keep.lin(u.pre_linear.weight), keep.ln(u.norm),
                        keep.ln(u.post_norm), keep.lin(u.post_linear.weight), ipre, ipost)


def collapse_read_out(read_out, p: int):
    """ConvTranspose2d(C,C,p,p) -> Conv1x1(C,C) -> Conv1x1(C,1) has no non-linearity in between
    (model/swin_gnn/swin_gnn.py:697-701): one [C] -> [p*p] affine map per token.  float64 on device."""
    W0 = read_out[0].weight.detach().double()              # [Cin, Cout, p, p]
    W1 = read_out[1].weight.detach().double()[:, :, 0, 0]  # [C1, C0]
    W2 = read_out[2].weight.detach().double()[0, :, 0, 0]  # [C1]
    v = W2 @ W1
    weff = torch.einsum("o,ioyx->iyx", v, W0).reshape(W0.shape[0], p * p)
    beff = v @ read_out[0].bias.detach().double() + W2 @ read_out[1].bias.detach().double() \
        + read_out[2].bias.detach().double()[0]
    return weff.float().contiguous(), float(beff)


def collapse_read_out_nodeadj(read_out, p: int):
    """NodeAdjSwinGNN read-out (model/swin_gnn/swin_gnn.py:900-904): ConvTranspose2d(C,C,p,p) -> Conv1x1(C,C) ->
    Conv1x1(C,C) with no non-linearity = one affine map [C] -> [p*p, C] per token.  float64 on device.
    Returns (W [p*p, C_out, C_in], b [C_out])."""
    W0 = read_out[0].weight.detach().double()              # [Cin, C0, p, p]
    W1 = read_out[1].weight.detach().double()[:, :, 0, 0]  # [C1, C0]
    W2 = read_out[2].weight.detach().double()[:, :, 0, 0]  # [C2, C1]
    M = W2 @ W1                                            # [C2, C0]
    weff = torch.einsum("oc,icyx->yxoi", M, W0).reshape(p * p, M.shape[0], W0.shape[0])
    beff = M @ read_out[0].bias.detach().double() + W2 @ read_out[1].bias.detach().double() \
        + read_out[2].bias.detach().double()
    return weff, beff


def nodeadj_desc(keep: Keep, model, p: int) -> L.SwgNodeAdj:
    C0 = model.embed_dim
    Cp = _kp(C0)
    weff, beff = collapse_read_out_nodeadj(model.read_out, p)
    wb = torch.zeros((p * p, Cp, Cp), dtype=torch.bfloat16, device=weff.device)
    wb[:, :C0, :C0] = weff
    bb = torch.zeros((p * p, Cp), dtype=torch.float32, device=weff.device)
    bb[:, :C0] = beff.float()
    keep(wb), keep(bb)
    keep.na_bufs = (wb, bb)
    ro = L.SwgLin(_ptr(wb), _ptr(bb), p * p * Cp, C0, Cp)
    am, nm = model.readout_adj_mlp, model.readout_node_mlp
    return L.SwgNodeAdj(model.out_chans_adj, model.out_chans_node, ro,
                        keep.lin(am.fc1.weight, am.fc1.bias), keep.lin(nm.fc1.weight, nm.fc1.bias),
                        keep.f32(am.fc2.weight), keep.f32(am.fc2.bias), keep.f32(nm.fc2.weight), keep.f32(nm.fc2.bias))


# tests of the host-only planning entry points (swg_*workspace_bytes) build descriptors over host memory; nothing can
# be launched from them (every launch needs device pointers and a GPU)
HOST_DESCRIPTORS_FOR_TESTS = False


class NetCache:
    """Descriptor + derived weights of one SwinGNN module on one device."""

    def __init__(self, model):
        keep = self.keep = Keep(registry=[])
        self.generation = 0
        dev = model.norm.weight.device
        if dev.type != "cuda" and not HOST_DESCRIPTORS_FOR_TESTS:
            raise RuntimeError("swingnn_amd runs on an MI355X only (module parameters are on %s); there is no "
                               "CPU fallback" % dev)
        self.device = dev
        C0 = model.embed_dim
        # conditioning vector layout: patch_embed first, then every block in execution order
        aff_w, aff_b = [model.patch_embed.affine.weight], [model.patch_embed.affine.bias]
        off = 2 * C0
        self.layer_objs = []
        layers = (L.SwgLayer * (2 * model.num_layers))()
        for li, layer in enumerate(list(model.down_layers) + list(model.up_layers)):
            blocks = (L.SwgBlock * len(layer.blocks))()
            for j, blk in enumerate(layer.blocks):
                blocks[j] = block_desc(keep, blk, off)
                aff_w.append(blk.affine.weight)
                aff_b.append(blk.affine.bias)
                off += 2 * blk.dim
            down = C.pointer(keep(merge_desc(keep, layer.downsample))) if layer.downsample is not None else None
            up = C.pointer(keep(breakup_desc(keep, layer.upsample))) if layer.upsample is not None else None
            keep(blocks)
            layers[li] = L.SwgLayer(len(layer.blocks), int(li >= model.num_layers),
                                    C.cast(blocks, C.POINTER(L.SwgBlock)),
                                    down if down is not None else C.POINTER(L.SwgMerge)(),
                                    up if up is not None else C.POINTER(L.SwgBreakup)())
        keep(layers)
        self.cond_total = off
        aw = keep(torch.cat([_f32(w) for w in aff_w], dim=0).contiguous())
        ab = keep(torch.cat([_f32(b) for b in aff_b], dim=0).contiguous())
        half = C0 // 2
        freqs = torch.arange(0, half, dtype=torch.float32, device=dev) / half
        freqs = keep(((1 / 10000) ** freqs).contiguous())  # PositionalEmbedding, swin_gnn.py:508-510
        p = model.patch_embed.patch_size[0]
        self.nodeadj = None
        if hasattr(model, "readout_adj_mlp"):  # NodeAdjSwinGNN: C0 -> C0 read-out + two MLP heads
            ro_w, ro_b = None, 0.0
            self.nodeadj = keep(nodeadj_desc(keep, model, p))
            self._na_wb, self._na_bb = keep.na_bufs
        else:
            ro_w, ro_b = collapse_read_out(model.read_out, p)
            keep(ro_w)
        pe = model.patch_embed
        pe_w = keep(_f32(pe.proj.weight).reshape(C0, -1).clone())
        self._aff_mods = [model.patch_embed.affine] + [blk.affine for layer in list(model.down_layers) + list(model.up_layers)
                                                      for blk in layer.blocks]
        self._aw, self._ab, self._pe_w, self._ro_w, self._p = aw, ab, pe_w, ro_w, p
        self.net = L.SwgNet(
            pe.img_size[0], p, pe.in_chans, C0, 2 * model.num_layers, int(model.self_condition), off, 0,
            _ptr(freqs), keep.f32(model.map_layer0.weight), keep.f32(model.map_layer0.bias),
            keep.f32(model.map_layer1.weight), keep.f32(model.map_layer1.bias),
            _ptr(aw), _ptr(ab), _ptr(pe_w), keep.f32(pe.proj.bias), keep.ln(pe.norm),
            C.cast(layers, C.POINTER(L.SwgLayer)), keep.ln(model.norm), _ptr(ro_w), ro_b)
        self.layers = layers
        self._ws_bytes = {}
        self._fused_ops = {}
        self.cond_rows = {}    # (mode, address of a batch-uniform label) -> precomputed conditioning row (register_cond_rows)
        self.table = RefreshTable(keep.registry, dev)

    def refresh(self, model):
        """Rewrite every derived operand in place from the current parameters (after an optimizer step): one
        swg_refresh_derived launch for the bf16 operands, a few torch copies for the small fp32 ones.  Pointers and
        descriptors stay valid."""
        self.cond_rows.clear()   # conditioning rows were computed from the old affine / map_layer weights
        self.table.run()
        build_rpb_tiles(self.keep)
        with torch.no_grad():
            torch.cat([m.weight for m in self._aff_mods], dim=0, out=self._aw)
            torch.cat([m.bias for m in self._aff_mods], dim=0, out=self._ab)
            self._pe_w.copy_(model.patch_embed.proj.weight.reshape(self._pe_w.shape))
            if self.nodeadj is not None:
                weff, beff = collapse_read_out_nodeadj(model.read_out, self._p)
                C0 = model.embed_dim
                self._na_wb[:, :C0, :C0] = weff
                self._na_bb[:, :C0] = beff.float()
            else:
                ro_w, ro_b = collapse_read_out(model.read_out, self._p)
                self._ro_w.copy_(ro_w)
                self.net.ro_b = ro_b
        self.generation += 1

    def fused_ops(self, B: int) -> int:
        """Ops of a pass over B graphs that take a batch-gated one-pass kernel (swg_fused_op_count)."""
        n = self._fused_ops.get(B)
        if n is None:
            n = self._fused_ops[B] = int(L.lib.swg_fused_op_count(C.byref(self.net), B))
        return n

    def ws_bytes(self, B: int) -> int:
        n = self._ws_bytes.get(B)
        if n is None:
            if self.nodeadj is not None:
                n = int(L.lib.swg_nodeadj_workspace_bytes(C.byref(self.net), C.byref(self.nodeadj), B))
            else:
                n = int(L.lib.swg_workspace_bytes(C.byref(self.net), B))
            if n == 0:
                raise L.SwgError("swg_workspace_bytes: unsupported network geometry")
            self._ws_bytes[B] = n
        return n

    def block(self, li: int, j: int) -> L.SwgBlock:
        return self.layers[li].blocks[j]


def params_key(model):
    # the module-tree walk of model.parameters() costs ~0.5 ms per call for these networks; the Parameter objects are
    # fixed after construction (load_state_dict / .to() / optimizers change them in place), so the list is kept
    ps = model.__dict__.get("_swg_plist")
    if ps is None:
        ps = model.__dict__["_swg_plist"] = list(model.parameters())
    return (tuple(p._version for p in ps), ps[0].data_ptr(), ps[-1].data_ptr(), len(ps), str(ps[0].device))


def get_cache(model) -> NetCache:
    """Derived operands of `model`: built once per parameter storage; when only the VALUES changed (version counters
    moved: optimizer step, load_state_dict) they are refreshed in place."""
    key = params_key(model)
    old = getattr(model, "_swg_key", None)
    if old != key:
        if old is not None and old[1:] == key[1:] and getattr(model, "_swg_cache", None) is not None:
            model._swg_cache.refresh(model)
        else:
            model._swg_cache = NetCache(model)
        model._swg_key = key
    return model._swg_cache


def _as_f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


PASS_COUNT = 0  # network passes launched through network_pass (reported by bench.py)

# A pass over a large batch is issued as SPLIT_STREAMS independent part-batches on side streams (graphs are independent;
# every kernel treats samples independently, so the results are bit-identical).  Within one stream kernel k+1 cannot
# start before the last workgroup of kernel k has drained (tile quantisation: e.g. 448 workgroups on 256 CUs leave a
# quarter of the second round idle); with two streams in flight the tail of one part's kernel overlaps the body of the
# other part's.  Measured on grid dim-96, B = 112: 7.57 -> 7.04 ms per pass (scripts/exp_two_streams.py); four parts are
# slower (more, smaller launches).  Off for small problems (launch-bound) and under stream capture.  While the
# library's per-kernel profiler is on, the SAME part-batch launches are issued one after the other on the caller's
# stream (overlapped kernels would be timed against each other): what is timed is what the product path launches.
SPLIT_STREAMS = int(os.environ.get("SWG_SPLIT_STREAMS", "2"))   # tuning knob for scripts/, not an API
# B * N * N below which a pass stays on one stream.  Measured (profiles/r03_split_sweep.txt, ms per pass, one stream ->
# two): grid B = 44 3.04 -> 2.88, B = 14 1.72 -> 1.74; DD B = 28 3.41 -> 3.28, B = 8 1.95 -> 1.99; the small-graph
# networks (N = 24, 14 blocks: many short launches) gain from 1 M: ego-small B = 2048 2.48 -> 2.32, B = 1024 1.86 -> 1.96.
SPLIT_MIN_WORK = int(os.environ.get("SWG_SPLIT_MIN_WORK", "4000000"))
SPLIT_SMALL_N, SPLIT_SMALL_DIV = 64, 4    # graphs of N <= 64 nodes: a quarter of that bound
_SIDE = {}
_PROFILING = False           # set by prof.profiled()


def launch_batch(B: int, N: int, cache=None) -> int:
    """Graphs per kernel launch of a pass over B graphs of N nodes (the part-batch when the pass is split).  With the
    NetCache given, a split that would push a stage below the token gate of a one-pass kernel (csrc/api.hip
    fused_gate: 12288 tokens; round 2 split grid B = 56 into 2 x 28 graphs = 16128 tokens at C = 384, under the
    16384 of its gate, and the stage fell back to the GEMM route) is refused: the part-batch must take the same
    kernels as the whole batch (swg_fused_op_count)."""
    parts = SPLIT_STREAMS
    if parts < 2 or B % parts or B * N * N < (SPLIT_MIN_WORK // SPLIT_SMALL_DIV if N <= SPLIT_SMALL_N else SPLIT_MIN_WORK):
        return B
    if cache is not None and cache.fused_ops(B // parts) != cache.fused_ops(B):
        return B
    return B // parts


def _side_streams(idx: int, n: int):
    key = (idx, n)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = [torch.cuda.Stream(device=idx) for _ in range(n)]
    return st


def _launch_pass(cache, B, mode, x, flags, labels, labels_uniform, sc, out, sym_err):
    ws = workspace(x.device, cache.ws_bytes(B))
    cond = cache.cond_rows.get((mode, labels.data_ptr())) if (labels_uniform and cache.cond_rows) else None
    L.check(L.lib.swg_precond_forward_cond(C.byref(cache.net), B, mode, x.data_ptr(), flags.data_ptr(),
                                           labels.data_ptr(), int(labels_uniform), _ptr(cond), _ptr(sc), out.data_ptr(),
                                           _ptr(sym_err), ws.data_ptr(), ws.numel(), stream_ptr()),
            "swg_precond_forward_cond")


def conditioning_table(cache, mode: int, labels: torch.Tensor) -> torch.Tensor:
    """[n, cond_total] conditioning rows of the n labels (swg_cond_table): what every pass with one batch-uniform
    label would compute at its head.  register_cond_rows() makes the passes use them."""
    labels = _as_f32(labels).reshape(-1)
    n = labels.numel()
    rows = torch.empty((n, cache.net.cond_total), dtype=torch.float32, device=labels.device)
    ws = torch.empty(4 * n * (5 + cache.net.C0 + 2 * 512) + 4096, dtype=torch.uint8, device=labels.device)
    L.check(L.lib.swg_cond_table(C.byref(cache.net), n, mode, labels.data_ptr(), rows.data_ptr(), ws.data_ptr(),
                                 ws.numel(), stream_ptr()), "swg_cond_table")
    return rows


def register_cond_rows(cache, mode: int, labels: torch.Tensor):
    """Precompute the conditioning rows of the batch-uniform labels a sampler is about to use (one per Heun step) and
    key them by the ADDRESS of each label: a pass whose (uniform) label tensor starts at labels[i] then skips its
    noise-embedding launches.  Returns a token for release_cond_rows().  The rows are tied to the current parameter
    values: NetCache.refresh() drops them."""
    labels = _as_f32(labels).reshape(-1)
    rows = conditioning_table(cache, mode, labels)
    keys = [(mode, labels.data_ptr() + 4 * i) for i in range(labels.numel())]
    for i, k in enumerate(keys):
        cache.cond_rows[k] = rows[i]
    return (labels, rows, keys)       # keeps the label vector (its addresses) and the table alive


def release_cond_rows(cache, token):
    for k in token[2]:
        cache.cond_rows.pop(k, None)


def network_pass(cache: NetCache, mode: int, x, node_flags, labels, labels_uniform: bool, self_cond, out=None,
                 sym_err=None):
    """One pass of the network through swg_precond_forward (mode 0: SwinGNN.forward, mode 1: Precond core)."""
    global PASS_COUNT
    PASS_COUNT += 1
    x = _as_f32(x)
    flags = _as_f32(node_flags)
    labels = _as_f32(labels)
    sc = None if self_cond is None else _as_f32(self_cond)
    B, N = x.shape[0], x.shape[-1]
    if x.dim() != 3 or x.shape[1] != N or N != cache.net.N:
        raise AssertionError(f"Input image size ({x.shape[-2]}*{N}) doesn't match model ({cache.net.N}*{cache.net.N}).")
    if out is None:
        out = torch.empty_like(x)
    h = launch_batch(B, N, cache)
    if h == B or (torch.cuda.is_current_stream_capturing() and not _CAPTURE_SPLIT):
        _launch_pass(cache, B, mode, x, flags, labels, labels_uniform, sc, out, sym_err)
        return out
    parts = B // h
    if _PROFILING:
        for lo in range(0, B, h):
            hi = lo + h
            _launch_pass(cache, h, mode, x[lo:hi], flags[lo:hi], labels if labels_uniform else labels[lo:hi],
                         labels_uniform, None if sc is None else sc[lo:hi], out[lo:hi], sym_err)
        return out
    dev_idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(dev_idx)
    fork = torch.cuda.Event()
    fork.record(cur)
    for i, side in enumerate(_side_streams(dev_idx, parts)):
        lo, hi = i * h, (i + 1) * h
        side.wait_event(fork)
        with torch.cuda.stream(side):
            _launch_pass(cache, h, mode, x[lo:hi], flags[lo:hi], labels if labels_uniform else labels[lo:hi],
                         labels_uniform, None if sc is None else sc[lo:hi], out[lo:hi], sym_err)
        join = torch.cuda.Event()
        join.record(side)
        cur.wait_event(join)   # later work on the caller's stream (and frees of x / out) is ordered after the parts
    return out


_CAPTURE_SPLIT = False       # set while graphed_pass records: the part-batch fork / join is captured too
GRAPH_SLOTS = 8              # captured passes kept per NetCache (each owns one [B,N,N] output buffer)
GRAPH_REPLAYS = 0            # passes executed as graph replays (reported by bench.py)


def graphed_pass(cache: NetCache, mode: int, x, node_flags, labels, labels_uniform: bool, self_cond, sym_err=None):
    """network_pass as ONE HIP-graph launch: the ~60 kernel launches of a pass (and the two part-batch branches with
    their fork / join) are recorded once per distinct set of buffer ADDRESSES and replayed afterwards, so the host
    thread enqueues one node instead of sixty (ranks sharing a contended host, small launch-bound problems).

    The recorded kernels read x / node_flags / labels / self_cond at the addresses they had when the graph was recorded
    and write into an output buffer the graph owns: callers keep their inputs in place and refill them between calls
    (EDMSampler(use_graphs=True) does), and the returned tensor is overwritten by the next replay of the same graph.
    Same kernels, same launch geometry, same order within a part-batch as the eager pass: results are bit-identical
    (tests/test_gpu_parity.py::test_graph_replayed_sampler_is_bit_identical).  Graphs live on the NetCache, i.e. they
    are dropped together with the weight images when the parameters are replaced."""
    global PASS_COUNT, GRAPH_REPLAYS, _CAPTURE_SPLIT, _WS_PRIVATE
    if x.dtype != torch.float32 or not x.is_contiguous() or _PROFILING or torch.cuda.is_current_stream_capturing():
        return network_pass(cache, mode, x, node_flags, labels, labels_uniform, self_cond, sym_err=sym_err)
    flags, labels = _as_f32(node_flags), _as_f32(labels)
    sc = None if self_cond is None else _as_f32(self_cond)
    key = (cache.generation, mode, tuple(x.shape), x.data_ptr(), flags.data_ptr(), labels.data_ptr(),
           bool(labels_uniform), _ptr(sc) or 0, _ptr(sym_err) or 0, torch.cuda.current_stream().cuda_stream)
    graphs = cache.__dict__.setdefault("_graphs", {})
    if graphs and next(iter(graphs))[0] != cache.generation:   # parameters refreshed: ro_b is baked into the nodes
        graphs.clear()
    ent = graphs.get(key)
    if ent is None:
        out = torch.empty_like(x)
        # eager run first: sets the kernels' LDS opt-ins outside the recording
        network_pass(cache, mode, x, flags, labels, labels_uniform, sc, out=out, sym_err=None)
        cs = _capture_stream(x.device)
        # Scratch of the recorded kernels: buffers the graph entry keeps alive (the eager pool of workspace() replaces
        # a buffer that a later, larger pass outgrows, which would leave the recorded addresses dangling).  Graphs
        # recorded from one caller stream replay in that stream's order, so they share one set; different caller
        # streams get different sets.  Allocated on the caller's stream: every replay joins back into it.
        B_, N_ = x.shape[0], x.shape[-1]
        dev_idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
        hb = launch_batch(B_, N_, cache)
        rec_streams = [cs] if hb == B_ else _side_streams(dev_idx, B_ // hb)
        pool = cache.__dict__.setdefault("_graph_ws", {})
        caller = torch.cuda.current_stream(dev_idx).cuda_stream
        private = {}
        for st in rec_streams:
            need = max(cache.ws_bytes(hb), 1 << 20)
            buf = pool.get((caller, st.cuda_stream))
            if buf is None or buf.numel() < need:
                buf = pool[(caller, st.cuda_stream)] = torch.empty(need, dtype=torch.uint8, device=x.device)
            private[(dev_idx, st.cuda_stream)] = buf
        graph = torch.cuda.CUDAGraph()
        _CAPTURE_SPLIT, _WS_PRIVATE = True, private
        try:
            with torch.cuda.graph(graph, stream=cs):
                network_pass(cache, mode, x, flags, labels, labels_uniform, sc, out=out, sym_err=sym_err)
        finally:
            _CAPTURE_SPLIT, _WS_PRIVATE = False, None
        PASS_COUNT -= 1          # the recording launched nothing
        while len(graphs) >= GRAPH_SLOTS:
            graphs.pop(next(iter(graphs)))
        # the recorded addresses (inputs, output, scratch) stay allocated for the life of the entry
        ent = graphs[key] = (graph, out, (x, flags, labels, sc, sym_err), tuple(private.values()))
    ent[0].replay()
    PASS_COUNT += 1
    GRAPH_REPLAYS += 1
    return ent[1]


_CAPTURE = {}


def _capture_stream(device):
    """Recording stream, one per (device, caller stream): graphs replayed from different streams never share the
    workspace of an unsplit pass (split passes use the part-batch streams' workspaces, like the eager path)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    st = _CAPTURE.get(key)
    if st is None:
        st = _CAPTURE[key] = torch.cuda.Stream(device=idx)
    return st


def nodeadj_pass(cache: NetCache, mode: int, adj, node, node_flags, labels, labels_uniform: bool, sc_adj, sc_node,
                 sym_err=None):
    """One pass of the molecule network through swg_nodeadj_forward (mode 0: NodeAdjSwinGNN.forward, mode 1:
    NodeAdjPrecond core).  adj [B,Ca,N,N], node [B,N,Cn] (already in the library's layout) -> (adj_out, node_out)."""
    na = cache.nodeadj
    adj, node = _as_f32(adj), _as_f32(node)
    flags = _as_f32(node_flags)
    labels = _as_f32(labels)
    sc_adj = None if sc_adj is None else _as_f32(sc_adj)
    sc_node = None if sc_node is None else _as_f32(sc_node)
    B, N = adj.shape[0], adj.shape[-1]
    if adj.dim() != 4 or adj.shape[1] != na.Ca or adj.shape[2] != N or N != cache.net.N:
        raise AssertionError(f"Input image size ({adj.shape[-2]}*{N}) doesn't match model ({cache.net.N}*{cache.net.N}).")
    if tuple(node.shape) != (B, N, na.Cn):
        raise AssertionError(f"node features {tuple(node.shape)} do not match [B, N, {na.Cn}]")
    for t, ref in ((sc_adj, adj), (sc_node, node)):
        if t is not None and t.shape != ref.shape:
            raise AssertionError("self-conditioning tensors must have the shapes of the inputs")
    out_adj, out_node = torch.empty_like(adj), torch.empty_like(node)
    ws = workspace(adj.device, cache.ws_bytes(B))
    L.check(L.lib.swg_nodeadj_forward(C.byref(cache.net), C.byref(na), B, mode, adj.data_ptr(), node.data_ptr(),
                                      flags.data_ptr(), labels.data_ptr(), int(labels_uniform), _ptr(sc_adj),
                                      _ptr(sc_node), out_adj.data_ptr(), out_node.data_ptr(), _ptr(sym_err),
                                      ws.data_ptr(), ws.numel(), stream_ptr()), "swg_nodeadj_forward")
    return out_adj, out_node
