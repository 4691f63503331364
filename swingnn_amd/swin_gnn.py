"""Drop-in ``SwinGNN`` for MI355X.

Same constructor keywords, ``forward`` signature and ``state_dict`` keys as the reference's
``model/swin_gnn/swin_gnn.py`` (SwinGNN :587-779), so ``utils/learning_utils.py:125-139`` can build
it unchanged and released checkpoints load with ``strict=True``.  The sub-modules exist to OWN the
parameters (nn.Linear / nn.LayerNorm / nn.Conv2d containers created in the reference's order, so
seeded initialisation draws the same numbers); all arithmetic runs in libswingnn_hip.so.
There is no PyTorch or CPU fallback: a missing library is an ImportError.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E

NOISE_EMB = L.SWG_NOISE_EMB


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


def _no_grad_path(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError("swingnn_amd: sub-module forwards are inference-only; gradients flow through "
                                  "SwinGNN.forward / Precond.forward (swingnn_amd/train.py)")


def _run(fn, what, *args):
    L.check(fn(*args), what)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (reference Mlp :9-25); executed fused with LN2 + residual by swg_ln_mlp_fwd."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class WindowAttention(nn.Module):
    """Parameter owner for W-MSA / SW-MSA (reference :60-139)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("swingnn_amd supports qk_scale=None and zero dropout (the shipped configs)")
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        w = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * w - 1) * (2 * w - 1), num_heads))
        iy, ix = torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")
        iy, ix = iy.reshape(-1), ix.reshape(-1)
        index = (iy[:, None] - iy[None, :] + w - 1) * (2 * w - 1) + (ix[:, None] - ix[None, :] + w - 1)
        self.register_buffer("relative_position_index", index)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    """Reference :158-277.  forward(x [B,L,C], emb [B,512]) -> [B,L,C] through swg_block_fwd (= swg_wattn_fwd + swg_ln_mlp_fwd)."""

    def __init__(self, dim, noise_emb_channels, input_resolution, num_heads, window_size=7, shift_size=0,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if drop or drop_path or mlp_ratio != 4. or act_layer is not nn.GELU or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("swingnn_amd supports mlp_ratio=4, GELU, LayerNorm and zero dropout / drop-path")
        if input_resolution[0] != input_resolution[1]:
            raise NotImplementedError("square token grids only (adjacency matrices)")
        self.dim, self.noise_emb_channels, self.input_resolution = dim, noise_emb_channels, input_resolution
        self.num_heads, self.mlp_ratio = num_heads, mlp_ratio
        self.window_size, self.shift_size = E.eff_window(min(input_resolution), window_size, shift_size)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.affine = nn.Linear(in_features=noise_emb_channels, out_features=dim * 2)
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=_pair(self.window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        attn_mask = None
        if self.shift_size > 0:
            # kept only for state_dict parity (strict loading); the kernel derives the mask from region ids
            res, w, s = input_resolution[0], self.window_size, self.shift_size
            v = torch.arange(res)
            reg = torch.where(v < res - w, 0, torch.where(v < res - s, 1, 2))
            ids = (3 * reg[:, None] + reg[None, :]).reshape(res // w, w, res // w, w).permute(0, 2, 1, 3)
            ids = ids.reshape(-1, w * w).to(torch.float32)
            diff = ids[:, None, :] - ids[:, :, None]
            attn_mask = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, x, emb):
        B, Ltok, Cdim = x.shape
        assert Ltok == self.input_resolution[0] * self.input_resolution[1], "input feature has wrong size"
        _no_grad_path(x, emb, self.affine.weight)
        keep = E.Keep()
        desc = E.block_desc(keep, self, 0)
        x = E._as_f32(x)
        cond = torch.addmm(self.affine.bias, E._as_f32(emb), self.affine.weight.t()).contiguous()  # tiny [B,2C] GEMV
        out = torch.empty_like(x)
        ws = E.workspace(x.device, 64 * x.numel() + (1 << 20))
        _run(L.lib.swg_block_fwd, "swg_block_fwd", C.byref(desc), B, x.data_ptr(), cond.data_ptr(), 2 * Cdim,
             out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr())
        return out


class PatchMerging(nn.Module):
    """Reference :298-335, through swg_patch_merge_fwd."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, Ltok, Cdim = x.shape
        assert Ltok == H * W, "input feature has wrong size"
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."
        _no_grad_path(x, self.reduction.weight)
        keep = E.Keep()
        desc = E.merge_desc(keep, self)
        x = E._as_f32(x)
        out = torch.empty((B, Ltok // 4, 2 * Cdim), dtype=torch.float32, device=x.device)
        ws = E.workspace(x.device, 16 * x.numel() + (1 << 20))
        _run(L.lib.swg_patch_merge_fwd, "swg_patch_merge_fwd", C.byref(desc), B, x.data_ptr(), out.data_ptr(),
             ws.data_ptr(), ws.numel(), E.stream_ptr())
        return out


class PatchBreakup(nn.Module):
    """Reference :347-403 (always built with skip_connection=True, :450).  ``forward(x)`` takes the
    already concatenated [B, L, dim] tensor like the reference; ``forward_pair(x, skip)`` is the
    fused form used by the network pass (concat of :743 folded into the first kernel)."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm, skip_connection=True):
        super().__init__()
        if not skip_connection:
            raise NotImplementedError("PatchBreakup without skip connection is never instantiated by SwinGNN")
        self.input_resolution, self.dim = input_resolution, dim
        self.pre_linear = nn.Linear(dim, dim, bias=False)
        self.norm = norm_layer(dim)
        self.true_dim = dim // 2
        self.post_linear = nn.Linear(self.true_dim // 2, self.true_dim // 2, bias=False)
        self.post_norm = norm_layer(self.true_dim // 2)

    def forward_pair(self, x, skip):
        H, W = self.input_resolution
        B, Ltok, half = x.shape
        assert Ltok == H * W, "input feature has wrong size"
        assert 2 * half == self.dim and skip.shape == x.shape
        _no_grad_path(x, skip, self.pre_linear.weight)
        keep = E.Keep()
        desc = E.breakup_desc(keep, self)
        x, skip = E._as_f32(x), E._as_f32(skip)
        out = torch.empty((B, 4 * Ltok, self.dim // 4), dtype=torch.float32, device=x.device)
        ws = E.workspace(x.device, 64 * x.numel() + (1 << 20))
        _run(L.lib.swg_patch_expand_fwd, "swg_patch_expand_fwd", C.byref(desc), B, x.data_ptr(), skip.data_ptr(),
             out.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr())
        return out

    def forward(self, x):
        assert x.shape[-1] == self.dim
        a, b = x.chunk(2, dim=-1)
        return self.forward_pair(a.contiguous(), b.contiguous())


class BasicLayer(nn.Module):
    """Reference :415-483: [upsample] -> blocks -> [downsample] (creation order kept for seeded init)."""

    def __init__(self, dim, noise_emb_channels, input_resolution, depth, num_heads, window_size,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 norm_layer=nn.LayerNorm, downsample=None, upsample=None, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        if upsample is not None:
            self.upsample = upsample(input_resolution, dim=dim * 4, norm_layer=norm_layer)
            input_resolution = tuple(item * 2 for item in input_resolution)
        else:
            self.upsample = None
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, noise_emb_channels=noise_emb_channels, input_resolution=input_resolution,
                                 num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x, emb):
        if self.upsample is not None:
            x = self.upsample(x)
        for blk in self.blocks:
            x = blk(x, emb)
        if self.downsample is not None:
            x = self.downsample(x)
        return x


class PositionalEmbedding(nn.Module):
    """Reference :500-513 (no parameters; evaluated inside swg_noise_embed)."""

    def __init__(self, num_channels, max_positions=10000, endpoint=False):
        super().__init__()
        if max_positions != 10000 or endpoint:
            raise NotImplementedError
        self.num_channels, self.max_positions, self.endpoint = num_channels, max_positions, endpoint


class PatchEmbed(nn.Module):
    """Reference :516-577, through swg_patch_embed_fwd."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, noise_emb_channels=512, norm_layer=None,
                 node_pe=False):
        super().__init__()
        if node_pe or norm_layer is None:
            raise NotImplementedError("swingnn_amd: PatchEmbed needs patch_norm=True and node_pe=False (shipped configs)")
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.patches_resolution = [self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.node_feat_pe_mlp = None
        self.affine = nn.Linear(in_features=noise_emb_channels, out_features=embed_dim * 2)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim)

    def forward(self, x, emb, node_flags=None):
        """x [B, in_chans, N, N] with in_chans 1 or 2 (2 = [self_cond, adjacency], :769-771) -> [B, L, C]."""
        B, cin, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        if cin != self.in_chans or cin not in (1, 2):
            raise NotImplementedError("swingnn_amd.PatchEmbed: 1 or 2 input channels")
        _no_grad_path(x, emb, self.affine.weight)
        keep = E.Keep()
        net = L.SwgNet()
        net.N, net.p, net.in_ch, net.C0, net.self_cond = H, self.patch_size[0], cin, self.embed_dim, int(cin == 2)
        net.pe_cond_off = 0
        net.pe_w = E._ptr(keep(E._f32(self.proj.weight).reshape(self.embed_dim, -1).contiguous()))
        net.pe_b = keep.f32(self.proj.bias)
        net.pe_ln = keep.ln(self.norm)
        x = E._as_f32(x)
        adj = keep(x[:, -1].contiguous())
        sc = keep(x[:, 0].contiguous()) if cin == 2 else None
        cond = torch.addmm(self.affine.bias, E._as_f32(emb), self.affine.weight.t()).contiguous()
        out = torch.empty((B, self.num_patches, self.embed_dim), dtype=torch.float32, device=x.device)
        _run(L.lib.swg_patch_embed_fwd, "swg_patch_embed_fwd", C.byref(net), B, adj.data_ptr(), E._ptr(sc), None, 0,
             cond.data_ptr(), 2 * self.embed_dim, out.data_ptr(), E.stream_ptr())
        return out


def _deepcopy_without_caches(self, memo):
    """copy.deepcopy of a denoiser (ema_pytorch.EMA deep-copies the model, utils/learning_utils.py:268) must not drag
    the derived-operand caches along (device descriptors, ctypes pointers): the copy builds its own on first use."""
    import copy
    new = self.__class__.__new__(self.__class__)
    memo[id(self)] = new
    for k, v in self.__dict__.items():
        if not k.startswith("_swg_"):
            new.__dict__[k] = copy.deepcopy(v, memo)
    return new


class SwinGNN(nn.Module):
    """Swin-UNet denoiser on N x N adjacency matrices (reference :587-779), executed by HIP kernels."""

    __deepcopy__ = _deepcopy_without_caches

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True,
                 use_checkpoint=False, self_condition=False, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("swingnn_amd: dropout / stochastic depth must be 0 (utils/learning_utils.py:137)")
        if in_chans != 1:
            raise NotImplementedError("swingnn_amd.SwinGNN: in_chans must be 1 (adjacency only)")
        self.num_layers = len(depths)
        self.embed_dim, self.patch_norm = embed_dim, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.self_condition = self_condition
        in_chans = in_chans * 2 if self_condition else in_chans
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      noise_emb_channels=NOISE_EMB, norm_layer=norm_layer if patch_norm else None)
        pr = self.patches_resolution = self.patch_embed.patches_resolution
        self.pos_drop = nn.Dropout(p=drop_rate)
        S = self.num_layers
        common = dict(noise_emb_channels=NOISE_EMB, window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=0., norm_layer=norm_layer,
                      use_checkpoint=use_checkpoint)
        self.down_layers = nn.ModuleList()
        for i in range(S):
            self.down_layers.append(BasicLayer(dim=int(embed_dim * 2 ** i),
                                               input_resolution=(pr[0] // 2 ** i, pr[1] // 2 ** i), depth=depths[i],
                                               num_heads=num_heads[i],
                                               downsample=PatchMerging if i < S - 1 else None, upsample=None, **common))
        self.up_layers = nn.ModuleList()
        for i in range(S):
            rest = S - i - 1
            k = rest if i == 0 else rest + 1  # layers with an upsample are built at the pre-upsample resolution
            self.up_layers.append(BasicLayer(dim=int(embed_dim * 2 ** rest),
                                             input_resolution=(pr[0] // 2 ** k, pr[1] // 2 ** k), depth=depths[rest],
                                             num_heads=num_heads[rest], downsample=None,
                                             upsample=PatchBreakup if i > 0 else None, **common))
        self.read_out = nn.Sequential(
            nn.ConvTranspose2d(embed_dim, embed_dim, kernel_size=patch_size, stride=patch_size),
            nn.Conv2d(embed_dim, embed_dim, kernel_size=1, stride=1, padding=0),
            nn.Conv2d(embed_dim, 1, kernel_size=1, stride=1, padding=0))
        self.map_noise = PositionalEmbedding(num_channels=embed_dim)
        self.map_layer0 = nn.Linear(in_features=embed_dim, out_features=NOISE_EMB)
        self.map_layer1 = nn.Linear(in_features=NOISE_EMB, out_features=NOISE_EMB)
        self.norm = norm_layer(self.embed_dim)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def forward(self, x, node_flags, noise_labels, self_cond=None):
        """x [B,N,N] (or [B,1,N,N]), node_flags [B,N], noise_labels [B], self_cond [B,N,N] | None -> [B,N,N]."""
        if x.dim() == 4:
            x = x.squeeze(1)
        if self_cond is not None and self_cond.dim() == 4:
            self_cond = self_cond.squeeze(1)
        if not self.self_condition:
            self_cond = None
        cache = E.get_cache(self)
        labels = noise_labels.reshape(-1)
        if labels.numel() == 1 and x.shape[0] != 1:
            labels = labels.expand(x.shape[0])
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .train import denoise_with_grad
            return denoise_with_grad(self, 0, x, node_flags, labels, self_cond)
        return E.network_pass(cache, 0, x, node_flags, labels, False, self_cond)


class NodeAdjSwinGNN(nn.Module):
    """Swin-UNet denoiser on node features + (multi-channel) adjacency matrices: the molecule model of the reference
    (model/swin_gnn/swin_gnn.py:782-1021), same constructor keywords, ``forward`` signature and ``state_dict`` keys,
    executed by swg_nodeadj_forward.  ``in_chans`` = 2 * node channels + adjacency channels
    (utils/learning_utils.py:184-214), doubled internally under self-conditioning (:825-826)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, out_chans_adj=1, out_chans_node=1,
                 norm_layer=nn.LayerNorm, patch_norm=True, use_checkpoint=False, self_condition=False,
                 symmetric_noise=True, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("swingnn_amd: dropout / stochastic depth must be 0 (utils/learning_utils.py:116)")
        if not symmetric_noise:
            raise NotImplementedError("swingnn_amd.NodeAdjSwinGNN: symmetric_noise=True (utils/learning_utils.py:119)")
        if in_chans != 2 * out_chans_node + out_chans_adj:
            raise NotImplementedError("swingnn_amd.NodeAdjSwinGNN: in_chans must be 2*out_chans_node + out_chans_adj "
                                      "(utils/learning_utils.py:203-212)")
        self.num_layers = len(depths)
        self.embed_dim, self.patch_norm = embed_dim, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.self_condition = self_condition
        in_chans = in_chans * 2 if self_condition else in_chans
        self.symmetric_noise = symmetric_noise
        self.out_chans_adj, self.out_chans_node = out_chans_adj, out_chans_node
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      noise_emb_channels=NOISE_EMB, norm_layer=norm_layer if patch_norm else None)
        pr = self.patches_resolution = self.patch_embed.patches_resolution
        self.pos_drop = nn.Dropout(p=drop_rate)
        S = self.num_layers
        common = dict(noise_emb_channels=NOISE_EMB, window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=0., norm_layer=norm_layer,
                      use_checkpoint=use_checkpoint)
        self.down_layers = nn.ModuleList()
        for i in range(S):
            self.down_layers.append(BasicLayer(dim=int(embed_dim * 2 ** i),
                                               input_resolution=(pr[0] // 2 ** i, pr[1] // 2 ** i), depth=depths[i],
                                               num_heads=num_heads[i],
                                               downsample=PatchMerging if i < S - 1 else None, upsample=None, **common))
        self.up_layers = nn.ModuleList()
        for i in range(S):
            rest = S - i - 1
            k = rest if i == 0 else rest + 1
            self.up_layers.append(BasicLayer(dim=int(embed_dim * 2 ** rest),
                                             input_resolution=(pr[0] // 2 ** k, pr[1] // 2 ** k), depth=depths[rest],
                                             num_heads=num_heads[rest], downsample=None,
                                             upsample=PatchBreakup if i > 0 else None, **common))
        self.read_out = nn.Sequential(
            nn.ConvTranspose2d(embed_dim, embed_dim, kernel_size=patch_size, stride=patch_size),
            nn.Conv2d(embed_dim, embed_dim, kernel_size=1, stride=1, padding=0),
            nn.Conv2d(embed_dim, embed_dim, kernel_size=1, stride=1, padding=0))
        self.map_noise = PositionalEmbedding(num_channels=embed_dim)
        self.map_layer0 = nn.Linear(in_features=embed_dim, out_features=NOISE_EMB)
        self.map_layer1 = nn.Linear(in_features=NOISE_EMB, out_features=NOISE_EMB)
        self.norm = norm_layer(self.embed_dim)
        self.readout_adj_mlp = Mlp(in_features=embed_dim, hidden_features=embed_dim, out_features=out_chans_adj)
        self.readout_node_mlp = Mlp(in_features=embed_dim, hidden_features=embed_dim, out_features=out_chans_node)
        self.apply(self._init_weights)

    _init_weights = SwinGNN._init_weights
    __deepcopy__ = _deepcopy_without_caches

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    # layout between the reference's tensors and the library's [B,Ca,N,N] / [B,N,Cn]
    def _adj_in(self, t):
        return None if t is None else (t.unsqueeze(1) if t.dim() == 3 else t)

    def _node_in(self, t):
        return None if t is None else (t.unsqueeze(-1) if t.dim() == 2 else t)

    def _adj_out(self, t):
        return t.squeeze(1) if self.out_chans_adj == 1 else t

    def _node_out(self, t):
        return t.squeeze(-1) if self.out_chans_node == 1 else t

    def forward(self, adj, node, node_flags, noise_labels, self_cond_x=None, self_cond_feat=None):
        """adj [B,N,N] | [B,Ca,N,N], node [B,N] | [B,N,Cn], node_flags [B,N], noise_labels [B] -> (adj_out, node_out)
        in the layouts of the inputs (reference :960-1021)."""
        if not self.self_condition:
            self_cond_x = self_cond_feat = None
        cache = E.get_cache(self)
        a = self._adj_in(adj)
        labels = noise_labels.reshape(-1)
        if labels.numel() == 1 and a.shape[0] != 1:
            labels = labels.expand(a.shape[0])
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .train import denoise_nodeadj_with_grad
            oa, on = denoise_nodeadj_with_grad(self, 0, a, self._node_in(node.to(torch.float32)), node_flags, labels,
                                               self._adj_in(self_cond_x), self._node_in(self_cond_feat))
            return self._adj_out(oa), self._node_out(on)
        oa, on = E.nodeadj_pass(cache, 0, a, self._node_in(node.to(torch.float32)), node_flags, labels, False,
                                self._adj_in(self_cond_x), self._node_in(self_cond_feat))
        return self._adj_out(oa), self._node_out(on)
