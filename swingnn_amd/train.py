"""Training path of the drop-in denoiser: forward that keeps op inputs, backward through the
``swg_*_bwd`` entry points of libswingnn_hip.so, wrapped in one ``torch.autograd.Function`` over the whole
``Precond(SwinGNN)`` / ``SwinGNN`` call so that ``loss.backward()`` (runner/trainer/trainer.py:135) fills ``.grad``
of the ordinary nn.Parameters (Adam, EMA, DDP and ``clip_grad_norm_`` see nothing unusual).

What is kept between forward and backward: the input of every op (block input, block mid-point, merge / breakup
inputs, final tokens) in fp32 -- each backward op recomputes its own intermediates.  Every token- or pixel-sized
tensor is produced by HIP kernels.  Two parameter-sized chains run as torch ops on the device because they are
smaller than a kernel launch is worth: the [B, 512] noise-embedding MLP + affine layers (reference :757-760, :195,
:554; its input gradient ``dcond`` comes from the kernels) and the chain rule through the read-out collapse
(three convolutions with no non-linearity, engine.collapse_read_out).  The gradient of a skip tensor used twice
(encoder successor + decoder concat, :743) is summed with one ``add_`` as autograd itself would.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib as L
from . import engine as E


def _kp(k):
    return (k + 31) // 32 * 32


def _wt(weight):
    """bf16 [K, Np]: transposed weight, N zero-padded to a multiple of 32 (operand of dX = dY W)."""
    w = weight.detach()
    N, K = w.shape
    out = torch.zeros((K, _kp(N)), dtype=torch.bfloat16, device=w.device)
    out[:, :N] = w.t()
    return out


def _padded_bf16(w2d):
    N, K = w2d.shape
    out = torch.zeros((N, _kp(K)), dtype=torch.bfloat16, device=w2d.device)
    out[:, :K] = w2d
    return out


class TrainCache:
    """Derived operands of the backward pass (transposed bf16 weights), built once per parameter version."""

    def __init__(self, model, cache: E.NetCache):
        self.model = model
        self.generation = cache.generation
        self.wt = {}
        registry = []
        for name, p in model.named_parameters():
            if p.dim() == 2 and name.endswith("weight") and ".affine." not in name and "map_layer" not in name:
                self.wt[name] = _wt(p)
                registry.append(("padT", (p,), self.wt[name], None))
        # fused front half of the MLP backward (csrc/mlp_bwd_front.hip): one weight image per narrow block
        self.mlp_bwd_img = {}
        keep = E.Keep(registry)
        self._keep = keep
        for lname, layer in [(f"down_layers.{i}", l) for i, l in enumerate(model.down_layers)] + \
                            [(f"up_layers.{i}", l) for i, l in enumerate(model.up_layers)]:
            for j, blk in enumerate(layer.blocks):
                if blk.dim in E.MLP_BWD_FRONT_DIMS:
                    self.mlp_bwd_img[f"{lname}.blocks.{j}"] = keep.image(E.mlp_bwd_front_image,
                                                                        (blk.mlp.fc1.weight, blk.mlp.fc2.weight))
        pe = model.patch_embed
        self.pe_wb = _padded_bf16(pe.proj.weight.detach().reshape(model.embed_dim, -1))
        registry.append(("pad", (pe.proj.weight,), self.pe_wb, None))
        self.nodeadj = cache.nodeadj is not None
        if self.nodeadj:  # collapsed read-out map [p*p*Cp, Cp] transposed: operand of dxn = drep Wro
            p_, C0 = pe.patch_size[0], model.embed_dim
            weff, _ = E.collapse_read_out_nodeadj(model.read_out, p_)
            Cp = _kp(C0)
            wt = torch.zeros((C0, p_ * p_, Cp), dtype=torch.bfloat16, device=weff.device)
            wt[:, :, :C0] = weff.permute(2, 0, 1)          # [Cin, ij, Cout]
            self.ro_wt = wt.reshape(C0, p_ * p_ * Cp).contiguous()
        else:
            ro_w, _ = E.collapse_read_out(model.read_out, pe.patch_size[0])
            self.ro_wb = _padded_bf16(ro_w)
        self._ws = {}
        self.table = E.RefreshTable(registry, pe.proj.weight.device)

    def refresh(self, cache):
        """Rewrite the transposed / padded operands in place after the parameters changed."""
        model = self.model
        self.table.run()
        with torch.no_grad():
            p_ = model.patch_embed.patch_size[0]
            if self.nodeadj:
                C0 = model.embed_dim
                weff, _ = E.collapse_read_out_nodeadj(model.read_out, p_)
                self.ro_wt.view(C0, p_ * p_, -1)[:, :, :C0] = weff.permute(2, 0, 1)
            else:
                ro_w, _ = E.collapse_read_out(model.read_out, p_)
                self.ro_wb[:, :ro_w.shape[1]] = ro_w
        self.generation = cache.generation

    def ws_bytes(self, cache, B):
        n = self._ws.get(B)
        if n is None:
            if self.nodeadj:
                n = int(L.lib.swg_nodeadj_train_workspace_bytes(C.byref(cache.net), C.byref(cache.nodeadj), B))
            else:
                n = int(L.lib.swg_train_workspace_bytes(C.byref(cache.net), B))
            if n == 0:
                raise L.SwgError("swg_train_workspace_bytes: unsupported network geometry")
            self._ws[B] = n
        return n


def get_train_cache(model, cache):
    if getattr(model, "_swg_train_key", None) is not cache:
        model._swg_train_cache = TrainCache(model, cache)
        model._swg_train_key = cache
    elif model._swg_train_cache.generation != cache.generation:
        model._swg_train_cache.refresh(cache)
    return model._swg_train_cache


class GradBook:
    """One zero-filled fp32 buffer holding the gradient of every kernel-side parameter, 256-byte aligned views."""

    def __init__(self, model, device):
        self.views = {}
        off = 0
        spans = []
        for name, p in model.named_parameters():
            spans.append((name, off, p.shape))
            off += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(off + 64, dtype=torch.float32, device=device)
        self.offset = {name: o for name, o, _ in spans}
        for name, o, shape in spans:
            n = 1
            for d in shape:
                n *= d
            self.views[name] = self.flat[o:o + n].view(shape)

    def ptr(self, name):
        return self.views[name].data_ptr()


class _Readiness:
    """Which tail of the flat gradient buffer is complete, derived from the modules whose backward has been enqueued
    (not from an assumed registration order): `flat[lo:]` is handed to the gradient sync only when EVERY parameter at
    or after `lo` belongs to a finished module.  `assert_complete` fails loudly for a model layout with a parameter no
    backward call owns."""

    def __init__(self, book, sync, model):
        self.sync = sync
        index = getattr(model, "_swg_ready_index", None)
        if index is None or index[0] != len(book.offset):  # static per model: built once
            entries = sorted((o, name) for name, o in book.offset.items())
            by_prefix = {}
            for i, (_, name) in enumerate(entries):
                parts = name.split(".")
                for d in range(1, len(parts) + 1):
                    by_prefix.setdefault(".".join(parts[:d]), []).append(i)
            index = model._swg_ready_index = (len(entries), entries, by_prefix)
        _, self.entries, self.by_prefix = index
        self.done = [False] * len(self.entries)
        self.tail = len(self.entries)  # entries[tail:] are all done

    def finish(self, *prefixes):
        for p in prefixes:
            for i in self.by_prefix.get(p, ()):
                self.done[i] = True
        moved = False
        while self.tail > 0 and self.done[self.tail - 1]:
            self.tail -= 1
            moved = True
        if moved and self.sync is not None:
            self.sync.ready_from(self.entries[self.tail][0])

    def assert_complete(self):
        missing = [name for (_, name), d in zip(self.entries, self.done) if not d]
        assert not missing, f"no backward call produced the gradient of {missing[:4]} (+{max(0, len(missing) - 4)} more)"


def _lin_train(tc, book, prefix, bias=True):
    return L.SwgLinTrain(tc.wt[prefix + ".weight"].data_ptr(), book.ptr(prefix + ".weight"),
                         book.ptr(prefix + ".bias") if bias else None)


def _ln_train(book, prefix):
    return L.SwgLnTrain(book.ptr(prefix + ".weight"), book.ptr(prefix + ".bias"))


def _layer_names(model):
    S = model.num_layers
    return [f"down_layers.{i}" for i in range(S)] + [f"up_layers.{i}" for i in range(S)]


def _run(fn, what, *args):
    L.check(fn(*args), what)


def _c(ptr_struct):
    return C.byref(ptr_struct)


def _trunk_forward(cache, B, tok, cond, ct, wp, wn, st):
    """Encoder / decoder layers (model/swin_gnn/swin_gnn.py:735-746) through the op-level entry points, keeping every
    op's input.  -> (final tokens, saved)"""
    net = cache.net
    f32 = dict(dtype=torch.float32, device=tok.device)
    saved, skips = [], {}
    cur = tok
    for li in range(net.n_layers):
        ly = cache.layers[li]
        if ly.up:
            up = ly.up.contents
            si = net.n_layers - 1 - li
            skip = skips[si]
            out = torch.empty((B, 4 * up.res * up.res, up.Cin // 4), **f32)
            _run(L.lib.swg_patch_expand_fwd, "swg_patch_expand_fwd", ly.up, B, cur.data_ptr(), skip.data_ptr(),
                 out.data_ptr(), wp, wn, st)
            saved.append(("up", li, si, cur, skip))
            cur = out
        for j in range(ly.n_blocks):
            blk = C.byref(ly.blocks[j])
            mid = torch.empty_like(cur)
            _run(L.lib.swg_wattn_fwd, "swg_wattn_fwd", blk, B, cur.data_ptr(), cond.data_ptr(), ct, mid.data_ptr(), wp, wn, st)
            out = torch.empty_like(cur)
            _run(L.lib.swg_ln_mlp_fwd, "swg_ln_mlp_fwd", blk, B, mid.data_ptr(), out.data_ptr(), wp, wn, st)
            saved.append(("blk", li, j, cur, mid))
            cur = out
        if ly.down:
            dn = ly.down.contents
            out = torch.empty((B, (dn.res // 2) ** 2, 2 * dn.C), **f32)
            _run(L.lib.swg_patch_merge_fwd, "swg_patch_merge_fwd", ly.down, B, cur.data_ptr(), out.data_ptr(), wp, wn, st)
            saved.append(("down", li, cur))
            skips[li] = out
            cur = out
    return cur, saved


def forward_keep(model, mode, x, node_flags, labels, self_cond, sym_err=None):
    """SwinGNN / Precond-core forward through the op-level entry points; returns (D, saved state for backward)."""
    cache = E.get_cache(model)
    tc = get_train_cache(model, cache)
    net = cache.net
    x = E._as_f32(x)
    flags = E._as_f32(node_flags)
    labels = E._as_f32(labels.reshape(-1))
    sc = None if self_cond is None else E._as_f32(self_cond)
    B, N = x.shape[0], x.shape[-1]
    if x.dim() != 3 or x.shape[1] != N or N != net.N:
        raise AssertionError(f"Input image size ({x.shape[-2]}*{N}) doesn't match model ({net.N}*{net.N}).")
    if labels.numel() != B:
        raise AssertionError("one noise label per sample is required in training")
    dev = x.device
    st = E.stream_ptr()
    ws = E.workspace(dev, max(cache.ws_bytes(B), tc.ws_bytes(cache, B)))
    wp, wn = ws.data_ptr(), ws.numel()
    f32 = dict(dtype=torch.float32, device=dev)
    coef = torch.empty((B, 4), **f32)
    cnoise = torch.empty((B,), **f32)
    _run(L.lib.swg_edm_coef, "swg_edm_coef", B, labels.data_ptr(), 0, mode, coef.data_ptr(), cnoise.data_ptr(), st)
    ct = net.cond_total
    cond = torch.empty((B, ct), **f32)
    _run(L.lib.swg_noise_embed, "swg_noise_embed", _c(net), B, cnoise.data_ptr(), cond.data_ptr(), wp, wn, st)
    hp = N // net.p
    tok = torch.empty((B, hp * hp, net.C0), **f32)
    _run(L.lib.swg_patch_embed_fwd, "swg_patch_embed_fwd", _c(net), B, x.data_ptr(), E._ptr(sc), coef.data_ptr() + 8, 4,
         cond.data_ptr(), ct, tok.data_ptr(), st)
    cur, saved = _trunk_forward(cache, B, tok, cond, ct, wp, wn, st)
    D = torch.empty_like(x)
    _run(L.lib.swg_readout_sym_precond_fwd, "swg_readout_sym_precond_fwd", _c(net), B, cur.data_ptr(), x.data_ptr(),
         flags.data_ptr(), coef.data_ptr(), D.data_ptr(), E._ptr(sym_err), wp, wn, st)
    state = dict(cache=cache, tc=tc, x=x, flags=flags, sc=sc, coef=coef, cnoise=cnoise, cond=cond, saved=saved,
                 tok_final=cur, B=B)
    return D, state


def _cond_chain_grads(model, cnoise, dcond):
    """torch chain rule through PositionalEmbedding -> map_layer0 -> silu -> map_layer1 -> silu -> affine layers."""
    C0 = model.embed_dim
    mods = [model.patch_embed.affine]
    for layer in list(model.down_layers) + list(model.up_layers):
        mods += [blk.affine for blk in layer.blocks]
    leaves = [model.map_layer0.weight, model.map_layer0.bias, model.map_layer1.weight, model.map_layer1.bias]
    for m in mods:
        leaves += [m.weight, m.bias]
    det = [p.detach().requires_grad_(True) for p in leaves]
    with torch.enable_grad():
        half = C0 // 2
        freqs = torch.arange(0, half, dtype=torch.float32, device=cnoise.device) / half
        freqs = (1 / 10000) ** freqs
        e = cnoise.detach().ger(freqs)
        e = torch.cat([e.cos(), e.sin()], dim=1)
        e = F.silu(F.linear(e, det[0], det[1]))
        e = F.silu(F.linear(e, det[2], det[3]))
        aw = torch.cat(det[4::2], dim=0)
        ab = torch.cat(det[5::2], dim=0)
        cond = F.linear(e, aw, ab)
    grads = torch.autograd.grad(cond, det, grad_outputs=dcond)
    return dict(zip([id(p) for p in leaves], grads))


def _readout_chain_grads(model, p, dweff, dbeff):
    """torch chain rule through engine.collapse_read_out (float64)."""
    ro = model.read_out
    leaves = [ro[0].weight, ro[0].bias, ro[1].weight, ro[1].bias, ro[2].weight, ro[2].bias]
    det = [t.detach().double().requires_grad_(True) for t in leaves]
    with torch.enable_grad():
        W0, b0, W1, b1, W2, b2 = det
        v = W2[0, :, 0, 0] @ W1[:, :, 0, 0]
        weff = torch.einsum("o,ioyx->iyx", v, W0).reshape(W0.shape[0], p * p)
        beff = v @ b0 + W2[0, :, 0, 0] @ b1 + b2[0]
    grads = torch.autograd.grad([weff, beff], det, grad_outputs=[dweff.double(), dbeff.double()])
    return dict(zip([id(t) for t in leaves], [g.float() for g in grads]))


def _trunk_backward(cache, tc, book, names, saved, dcur, cond, ct, dcond, B, wp, wn, st, ready):
    """Backward through the encoder / decoder layers: gradient of the final tokens -> gradient of the embedded tokens;
    parameter gradients accumulate into `book`, conditioning gradients into `dcond`."""
    dskips = {}
    for entry in reversed(saved):
        kind, li = entry[0], entry[1]
        ly = cache.layers[li]
        if kind == "down":
            x_in = entry[2]
            if li in dskips:
                dcur = dcur.add_(dskips.pop(li))
            pre = names[li] + ".downsample"
            tr = L.SwgMergeTrain(_ln_train(book, pre + ".norm"), _lin_train(tc, book, pre + ".reduction", bias=False))
            dx = torch.empty_like(x_in)
            _run(L.lib.swg_patch_merge_bwd, "swg_patch_merge_bwd", ly.down, _c(tr), B, x_in.data_ptr(), dcur.data_ptr(),
                 dx.data_ptr(), wp, wn, st)
            dcur = dx
        elif kind == "blk":
            j, x_in, mid = entry[2], entry[3], entry[4]
            pre = f"{names[li]}.blocks.{j}"
            tr = L.SwgBlockTrain(_ln_train(book, pre + ".norm1"), _ln_train(book, pre + ".norm2"),
                                 _lin_train(tc, book, pre + ".attn.qkv"), _lin_train(tc, book, pre + ".attn.proj"),
                                 _lin_train(tc, book, pre + ".mlp.fc1"), _lin_train(tc, book, pre + ".mlp.fc2"),
                                 book.ptr(pre + ".attn.relative_position_bias_table"),
                                 E._ptr(tc.mlp_bwd_img.get(pre)))
            blk = C.byref(ly.blocks[j])
            dmid = torch.empty_like(mid)
            _run(L.lib.swg_ln_mlp_bwd, "swg_ln_mlp_bwd", blk, _c(tr), B, mid.data_ptr(), dcur.data_ptr(), dmid.data_ptr(),
                 wp, wn, st)
            dx = torch.empty_like(x_in)
            _run(L.lib.swg_wattn_bwd, "swg_wattn_bwd", blk, _c(tr), B, x_in.data_ptr(), cond.data_ptr(), ct,
                 dmid.data_ptr(), dx.data_ptr(), dcond.data_ptr(), ct, wp, wn, st)
            dcur = dx
        else:  # "up"
            si, x_in, skip = entry[2], entry[3], entry[4]
            pre = names[li] + ".upsample"
            tr = L.SwgBreakupTrain(_lin_train(tc, book, pre + ".pre_linear", bias=False), _ln_train(book, pre + ".norm"),
                                   _ln_train(book, pre + ".post_norm"),
                                   _lin_train(tc, book, pre + ".post_linear", bias=False))
            dx, dskip = torch.empty_like(x_in), torch.empty_like(skip)
            _run(L.lib.swg_patch_expand_bwd, "swg_patch_expand_bwd", ly.up, _c(tr), B, x_in.data_ptr(), skip.data_ptr(),
                 dcur.data_ptr(), dx.data_ptr(), dskip.data_ptr(), wp, wn, st)
            dskips[si] = dskip
            dcur = dx
        ready.finish(pre)  # the kernels that write this module's gradients are enqueued
    return dcur


def backward_all(model, state, dD):
    """-> {id(param): grad} for every parameter of `model`."""
    cache, tc = state["cache"], state["tc"]
    net = cache.net
    B = state["B"]
    dev = dD.device
    dD = E._as_f32(dD)
    sync = getattr(model, "_swg_grad_sync", None)
    if sync is not None:  # data parallel: sum over ranks of (dD / W)-gradients = the DDP average
        dD = dD / sync.world_size
    st = E.stream_ptr()
    ws = E.workspace(dev, max(cache.ws_bytes(B), tc.ws_bytes(cache, B)))
    wp, wn = ws.data_ptr(), ws.numel()
    f32 = dict(dtype=torch.float32, device=dev)
    book = GradBook(model, dev)
    names = _layer_names(model)
    if sync is not None:
        sync.begin(book.flat)
    ready = _Readiness(book, sync, model)
    # flat slots of the parameters whose gradient comes from the parameter-sized torch chain rules below stay unused
    ready.finish("map_layer0", "map_layer1", "read_out")
    ct = net.cond_total
    cond = state["cond"]
    p = net.p
    dro_w = torch.zeros((net.C0, p * p), **f32)
    dro_b = torch.zeros((32,), **f32)
    ends = L.SwgEndsTrain(tc.pe_wb.data_ptr(), book.ptr("patch_embed.proj.weight"), book.ptr("patch_embed.proj.bias"),
                          _ln_train(book, "patch_embed.norm"), _ln_train(book, "norm"), tc.ro_wb.data_ptr(),
                          dro_w.data_ptr(), dro_b.data_ptr())
    tok_final = state["tok_final"]
    dcur = torch.empty_like(tok_final)
    _run(L.lib.swg_readout_sym_precond_bwd, "swg_readout_sym_precond_bwd", _c(net), _c(ends), B, tok_final.data_ptr(),
         state["flags"].data_ptr(), state["coef"].data_ptr(), dD.data_ptr(), dcur.data_ptr(), wp, wn, st)
    ready.finish("norm")
    dcond = torch.zeros((B, ct), **f32)
    dcur = _trunk_backward(cache, tc, book, names, state["saved"], dcur, cond, ct, dcond, B, wp, wn, st, ready)
    _run(L.lib.swg_patch_embed_bwd, "swg_patch_embed_bwd", _c(net), _c(ends), B, state["x"].data_ptr(),
         E._ptr(state["sc"]), state["coef"].data_ptr() + 8, 4, cond.data_ptr(), ct, dcur.data_ptr(), dcond.data_ptr(), ct,
         wp, wn, st)
    ready.finish("patch_embed")
    ready.assert_complete()
    out = {id(prm): book.views[name] for name, prm in model.named_parameters()}
    small = _cond_chain_grads(model, state["cnoise"], dcond)
    small.update(_readout_chain_grads(model, p, dro_w, dro_b[:p * p].sum()))
    if sync is not None:
        keys = list(small)
        vals = sync.finish([small[k].contiguous() for k in keys])
        small = dict(zip(keys, vals))
    out.update(small)
    return out


class DenoiserFn(torch.autograd.Function):
    """D = Precond core (mode 1) or SwinGNN.forward (mode 0) with gradients to the module parameters."""

    @staticmethod
    def forward(ctx, model, mode, x, node_flags, labels, self_cond, sym_err, *params):
        D, state = forward_keep(model, mode, x, node_flags, labels, self_cond, sym_err)
        ctx.model, ctx.state, ctx.params = model, state, params
        return D

    @staticmethod
    def backward(ctx, dD):
        grads = backward_all(ctx.model, ctx.state, dD.contiguous())
        ctx.state = None
        return (None,) * 7 + tuple(grads[id(p)] if p.requires_grad else None for p in ctx.params)


def denoise_with_grad(model, mode, x, node_flags, labels, self_cond, sym_err=None):
    params = tuple(model.parameters())
    return DenoiserFn.apply(model, mode, x, node_flags, labels, self_cond, sym_err, *params)


# ---- molecule path: NodeAdjPrecond(NodeAdjSwinGNN) (runner/trainer/trainer_node_adj.py:95-131) --------------------
def forward_keep_nodeadj(model, mode, adj, node, node_flags, labels, sc_adj, sc_node, sym_err=None):
    """adj [B,Ca,N,N], node [B,N,Cn] (library layout) -> ((D_adj, D_node), saved state)."""
    cache = E.get_cache(model)
    tc = get_train_cache(model, cache)
    net, na = cache.net, cache.nodeadj
    adj, node = E._as_f32(adj), E._as_f32(node)
    flags = E._as_f32(node_flags)
    labels = E._as_f32(labels.reshape(-1))
    sc_adj = None if sc_adj is None else E._as_f32(sc_adj)
    sc_node = None if sc_node is None else E._as_f32(sc_node)
    B, N = adj.shape[0], adj.shape[-1]
    if adj.dim() != 4 or adj.shape[1] != na.Ca or adj.shape[2] != N or N != net.N:
        raise AssertionError(f"Input image size ({adj.shape[-2]}*{N}) doesn't match model ({net.N}*{net.N}).")
    if tuple(node.shape) != (B, N, na.Cn):
        raise AssertionError(f"node features {tuple(node.shape)} do not match [B, N, {na.Cn}]")
    if labels.numel() != B:
        raise AssertionError("one noise label per sample is required in training")
    dev = adj.device
    st = E.stream_ptr()
    ws = E.workspace(dev, max(cache.ws_bytes(B), tc.ws_bytes(cache, B)))
    wp, wn = ws.data_ptr(), ws.numel()
    f32 = dict(dtype=torch.float32, device=dev)
    coef = torch.empty((B, 4), **f32)
    cnoise = torch.empty((B,), **f32)
    _run(L.lib.swg_edm_coef, "swg_edm_coef", B, labels.data_ptr(), 0, mode, coef.data_ptr(), cnoise.data_ptr(), st)
    ct = net.cond_total
    cond = torch.empty((B, ct), **f32)
    _run(L.lib.swg_noise_embed, "swg_noise_embed", _c(net), B, cnoise.data_ptr(), cond.data_ptr(), wp, wn, st)
    hp = N // net.p
    tok = torch.empty((B, hp * hp, net.C0), **f32)
    _run(L.lib.swg_nodeadj_embed_fwd, "swg_nodeadj_embed_fwd", _c(net), _c(na), B, adj.data_ptr(), node.data_ptr(),
         E._ptr(sc_adj), E._ptr(sc_node), flags.data_ptr(), coef.data_ptr() + 8, 4, cond.data_ptr(), ct, tok.data_ptr(), st)
    cur, saved = _trunk_forward(cache, B, tok, cond, ct, wp, wn, st)
    Da, Dn = torch.empty_like(adj), torch.empty_like(node)
    _run(L.lib.swg_nodeadj_readout_fwd, "swg_nodeadj_readout_fwd", _c(net), _c(na), B, cur.data_ptr(), adj.data_ptr(),
         node.data_ptr(), flags.data_ptr(), coef.data_ptr() if mode else None, Da.data_ptr(), Dn.data_ptr(),
         E._ptr(sym_err), wp, wn, st)
    state = dict(cache=cache, tc=tc, adj=adj, node=node, flags=flags, sc_adj=sc_adj, sc_node=sc_node, coef=coef,
                 cnoise=cnoise, cond=cond, saved=saved, tok_final=cur, B=B, mode=mode)
    return (Da, Dn), state


def _readout_chain_grads_nodeadj(model, p, dweff, dbeff):
    """torch chain rule through engine.collapse_read_out_nodeadj (float64): dweff [p*p, Cout, Cin], dbeff [Cout]."""
    ro = model.read_out
    leaves = [ro[0].weight, ro[0].bias, ro[1].weight, ro[1].bias, ro[2].weight, ro[2].bias]
    det = [t.detach().double().requires_grad_(True) for t in leaves]
    with torch.enable_grad():
        W0, b0, W1, b1, W2, b2 = det
        M = W2[:, :, 0, 0] @ W1[:, :, 0, 0]
        weff = torch.einsum("oc,icyx->yxoi", M, W0).reshape(p * p, M.shape[0], W0.shape[0])
        beff = M @ b0 + W2[:, :, 0, 0] @ b1 + b2
    grads = torch.autograd.grad([weff, beff], det, grad_outputs=[dweff.double(), dbeff.double()])
    return dict(zip([id(t) for t in leaves], [g.float() for g in grads]))


def backward_all_nodeadj(model, state, dDa, dDn):
    cache, tc = state["cache"], state["tc"]
    net, na = cache.net, cache.nodeadj
    B = state["B"]
    dev = dDa.device
    dDa, dDn = E._as_f32(dDa), E._as_f32(dDn)
    sync = getattr(model, "_swg_grad_sync", None)
    if sync is not None:
        dDa, dDn = dDa / sync.world_size, dDn / sync.world_size
    st = E.stream_ptr()
    ws = E.workspace(dev, max(cache.ws_bytes(B), tc.ws_bytes(cache, B)))
    wp, wn = ws.data_ptr(), ws.numel()
    f32 = dict(dtype=torch.float32, device=dev)
    book = GradBook(model, dev)
    names = _layer_names(model)
    if sync is not None:
        sync.begin(book.flat)
    ready = _Readiness(book, sync, model)
    ready.finish("map_layer0", "map_layer1", "read_out")
    ct = net.cond_total
    cond = state["cond"]
    p, C0 = net.p, net.C0
    Cp = _kp(C0)
    dro_w = torch.zeros((p * p * Cp, C0), **f32)
    dro_b = torch.zeros((p * p * Cp,), **f32)
    tr = L.SwgNodeAdjTrain(tc.ro_wt.data_ptr(), dro_w.data_ptr(), dro_b.data_ptr(),
                           _lin_train(tc, book, "readout_adj_mlp.fc1"), _lin_train(tc, book, "readout_node_mlp.fc1"),
                           book.ptr("readout_adj_mlp.fc2.weight"), book.ptr("readout_adj_mlp.fc2.bias"),
                           book.ptr("readout_node_mlp.fc2.weight"), book.ptr("readout_node_mlp.fc2.bias"),
                           _ln_train(book, "norm"))
    tok_final = state["tok_final"]
    dcur = torch.empty_like(tok_final)
    _run(L.lib.swg_nodeadj_readout_bwd, "swg_nodeadj_readout_bwd", _c(net), _c(na), _c(tr), B, tok_final.data_ptr(),
         state["flags"].data_ptr(), state["coef"].data_ptr() if state["mode"] else None, dDa.data_ptr(), dDn.data_ptr(),
         dcur.data_ptr(), wp, wn, st)
    ready.finish("norm", "readout_adj_mlp", "readout_node_mlp")
    dcond = torch.zeros((B, ct), **f32)
    dcur = _trunk_backward(cache, tc, book, names, state["saved"], dcur, cond, ct, dcond, B, wp, wn, st, ready)
    ends = L.SwgEndsTrain(tc.pe_wb.data_ptr(), book.ptr("patch_embed.proj.weight"), book.ptr("patch_embed.proj.bias"),
                          _ln_train(book, "patch_embed.norm"), _ln_train(book, "norm"), None, None, None)
    _run(L.lib.swg_nodeadj_embed_bwd, "swg_nodeadj_embed_bwd", _c(net), _c(na), _c(ends), B, state["adj"].data_ptr(),
         state["node"].data_ptr(), E._ptr(state["sc_adj"]), E._ptr(state["sc_node"]), state["flags"].data_ptr(),
         state["coef"].data_ptr() + 8, 4, cond.data_ptr(), ct, dcur.data_ptr(), dcond.data_ptr(), ct, wp, wn, st)
    ready.finish("patch_embed")
    ready.assert_complete()
    out = {id(prm): book.views[name] for name, prm in model.named_parameters()}
    small = _cond_chain_grads(model, state["cnoise"], dcond)
    dweff = dro_w.view(p * p, Cp, C0)[:, :C0, :]
    dbeff = dro_b.view(p * p, Cp)[:, :C0].sum(0)
    small.update(_readout_chain_grads_nodeadj(model, p, dweff, dbeff))
    if sync is not None:
        keys = list(small)
        vals = sync.finish([small[k].contiguous() for k in keys])
        small = dict(zip(keys, vals))
    out.update(small)
    return out


class NodeAdjDenoiserFn(torch.autograd.Function):
    """(D_adj, D_node) = NodeAdjPrecond core (mode 1) or NodeAdjSwinGNN.forward (mode 0) with parameter gradients."""

    @staticmethod
    def forward(ctx, model, mode, adj, node, node_flags, labels, sc_adj, sc_node, sym_err, *params):
        (Da, Dn), state = forward_keep_nodeadj(model, mode, adj, node, node_flags, labels, sc_adj, sc_node, sym_err)
        ctx.model, ctx.state, ctx.params = model, state, params
        return Da, Dn

    @staticmethod
    def backward(ctx, dDa, dDn):
        grads = backward_all_nodeadj(ctx.model, ctx.state, dDa.contiguous(), dDn.contiguous())
        ctx.state = None
        return (None,) * 9 + tuple(grads[id(p)] if p.requires_grad else None for p in ctx.params)


def denoise_nodeadj_with_grad(model, mode, adj, node, node_flags, labels, sc_adj, sc_node, sym_err=None):
    params = tuple(model.parameters())
    return NodeAdjDenoiserFn.apply(model, mode, adj, node, node_flags, labels, sc_adj, sc_node, sym_err, *params)
