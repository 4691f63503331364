"""CPU oracle: functional fp32 restatement of the SwinGNN denoiser hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED against
golden vectors generated from the reference modules (tests/golden/make_golden.py).

Every function is a pure function of a ``state_dict`` (the reference's key names) and a
geometry dict ``cfg``; there are no nn.Module classes here, so the state-dict key contract
is checked independently of the product's module tree.  Citations are ``file:line`` under
/root/reference.

cfg keys: img_size, patch_size, in_chans, embed_dim, depths, num_heads, window_size,
          self_condition  (same meaning as SwinGNN.__init__, model/swin_gnn/swin_gnn.py:610-617)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SIGMA_DATA = 0.5          # runner/objectives/edm.py:60-63
NOISE_EMB = 512           # model/swin_gnn/swin_gnn.py:634
MASK_NEG = -100.0         # model/swin_gnn/swin_gnn.py:226

# --------------------------------------------------------------------------------------
# bf16-operand mode (test infrastructure for the tolerance tests, NOT the reference's arithmetic)
# --------------------------------------------------------------------------------------
# The reference computes in fp32 throughout.  The HIP path rounds the OPERANDS of the token-wise matrix products to
# bfloat16 (fp32 accumulate; everything else fp32).  With `bf16_operands()` active the oracle rounds the same
# operands at the same places -- the inputs and weights of qkv / proj / fc1 / fc2 / PatchMerging.reduction /
# PatchBreakup.pre_linear / post_linear, q (scaled, with bias), k, v, the unnormalised softmax numerators and the
# attention output -- so that what is left between the two is accumulation order and the transcendental
# approximations.  The parity tests then hold the unit-gain ('stress') regime to a bar ~20x tighter than against the
# fp32 goldens, where the bar has to absorb the rounding itself.  Patch embedding, read-out, noise embedding and the
# conditioning GEMV are fp32-grade in the HIP path and stay fp32 here.  Under autograd the casts round the incoming
# gradients at the same points (the HIP backward feeds bf16 dY operands to its products too).
_BF16 = False
_BF16_Q = "auto"
LOG2E = 1.4426950408889634


class bf16_operands:
    """Context manager: round matrix-product operands to bfloat16 like the HIP kernels do.  `q`: where the query is
    rounded -- "log2" (fused window kernels: q * hd^-1/2 * log2 e), "raw" (GEMM route: the unscaled qkv row) or
    "auto" (per block, by the geometry rule of csrc/api.hip: fused for head_dim 32, C in {96, 192, 384}, windows
    4 / 6 / 8 at C = 96 and 2 / 4 / 6 / 8 above; head_dim 20 at C = 60 / 120 / 240 with windows 2 / 4 / 8).  At
    unit-gain weights the softmax is sharp enough that this choice alone moves a block's output by 3e-3 of its scale."""

    def __init__(self, q="auto"):
        self.q = q

    def __enter__(self):
        global _BF16, _BF16_Q
        self._old, _BF16, _BF16_Q = (_BF16, _BF16_Q), True, self.q
        return self

    def __exit__(self, *exc):
        global _BF16, _BF16_Q
        _BF16, _BF16_Q = self._old
        return False


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32) if _BF16 else t


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
def eff_window(res: int, window: int, shift: int):
    """Window clamp rule, model/swin_gnn/swin_gnn.py:189-192."""
    if res <= window:
        return res, 0
    return window, shift


def stages(cfg):
    """List of layer descriptors in execution order (encoder then decoder).

    model/swin_gnn/swin_gnn.py:649-694.  Each: dict(prefix, dim, res (block resolution),
    depth, heads, up (bool), down (bool)).
    """
    S = len(cfg["depths"])
    r0 = cfg["img_size"] // cfg["patch_size"]
    out = []
    for i in range(S):
        out.append(dict(prefix=f"down_layers.{i}", dim=cfg["embed_dim"] * 2 ** i, res=r0 // 2 ** i,
                        depth=cfg["depths"][i], heads=cfg["num_heads"][i], up=False, down=i < S - 1))
    for i in range(S):
        rest = S - 1 - i
        # blocks run at the up-sampled resolution r0 / 2**rest (swin_gnn.py:448-451, 672-679)
        out.append(dict(prefix=f"up_layers.{i}", dim=cfg["embed_dim"] * 2 ** rest, res=r0 // 2 ** rest,
                        depth=cfg["depths"][rest], heads=cfg["num_heads"][rest], up=i > 0, down=False))
    return out


def relative_position_index(w: int) -> torch.Tensor:
    """index[i,j] = (iy-jy+w-1)(2w-1) + (ix-jx+w-1); model/swin_gnn/swin_gnn.py:88-97."""
    iy, ix = torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")
    iy, ix = iy.reshape(-1), ix.reshape(-1)
    return (iy[:, None] - iy[None, :] + w - 1) * (2 * w - 1) + (ix[:, None] - ix[None, :] + w - 1)


def shift_mask(res: int, w: int, s: int) -> torch.Tensor:
    """[nW, n, n] 0 / -100 mask of a shifted block; model/swin_gnn/swin_gnn.py:207-226."""
    def region(v):  # region id along one axis, in the shifted frame
        return torch.where(v < res - w, 0, torch.where(v < res - s, 1, 2))
    coords = torch.arange(res)
    ids = 3 * region(coords)[:, None] + region(coords)[None, :]            # [res, res]
    ids = ids.reshape(res // w, w, res // w, w).permute(0, 2, 1, 3).reshape(-1, w * w)
    diff = ids[:, None, :] - ids[:, :, None]
    return torch.where(diff != 0, torch.tensor(MASK_NEG), torch.tensor(0.0))


# --------------------------------------------------------------------------------------
# tensor helpers (utils/graph_utils.py)
# --------------------------------------------------------------------------------------
def mask_adjs(a: torch.Tensor, flags: torch.Tensor) -> torch.Tensor:
    """Zero rows and columns whose flag is 0; out-of-place.  utils/graph_utils.py:5-38."""
    keep = flags.to(torch.bool)
    if a.dim() == 4:
        keep = keep[:, None]
    return a * (keep[..., :, None] & keep[..., None, :]).to(a.dtype)


def sym_noise_from_raw(raw: torch.Tensor) -> torch.Tensor:
    """triu(raw, 1) + transpose; utils/graph_utils.py:113-119 with the randn draw factored out."""
    n = raw.triu(1)
    return n + n.transpose(-1, -2)


def symmetry_residual(a: torch.Tensor) -> torch.Tensor:
    """sum |a - a^T|, the quantity asserted < 1e-2 in utils/graph_utils.py:89-94."""
    return (a - a.transpose(-1, -2)).abs().sum()


# --------------------------------------------------------------------------------------
# network pieces (model/swin_gnn/swin_gnn.py)
# --------------------------------------------------------------------------------------
def _lin(x, sd, pfx):
    return F.linear(x, sd[pfx + ".weight"], sd.get(pfx + ".bias"))


def _lin_mm(x, sd, pfx):
    """A Linear that the HIP path runs on MFMA: operands rounded in bf16-operand mode, bias added in fp32."""
    return F.linear(_rb(x), _rb(sd[pfx + ".weight"]), sd.get(pfx + ".bias"))


def _ln(x, sd, pfx):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + ".weight"], sd[pfx + ".bias"], 1e-5)


def _cond(x, emb, sd, pfx):
    """x <- silu(shift + x * (scale + 1)); swin_gnn.py:238-240 / 574-576."""
    scale, shift = _lin(emb, sd, pfx)[:, None, :].chunk(2, dim=-1)
    return F.silu(shift + x * (scale + 1))


def noise_embedding(c_noise, sd, C):
    """swin_gnn.py:507-513 and 757-760."""
    freqs = torch.arange(0, C // 2, dtype=torch.float32) / (C // 2)
    freqs = (1 / 10000) ** freqs
    ang = c_noise.to(torch.float32).ger(freqs)
    e = torch.cat([ang.cos(), ang.sin()], dim=1)
    e = F.silu(_lin(e, sd, "map_layer0"))
    return F.silu(_lin(e, sd, "map_layer1"))


def window_attention(xw, sd, pfx, heads, w, mask):
    """xw [B_, n, C] -> [B_, n, C]; swin_gnn.py:108-139."""
    B_, n, C = xw.shape
    hd = C // heads
    qkv = _lin_mm(xw, sd, pfx + ".qkv").reshape(B_, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q_mode = _BF16_Q
    if q_mode == "auto":   # which route the library takes for this geometry (csrc/api.hip wattn_impl)
        fused = (hd == 32 and C in (96, 192, 384) and w in ((4, 6, 8) if C == 96 else (2, 4, 6, 8))
                 or hd == 20 and C in (60, 120, 240) and w in (2, 4, 8))
        q_mode = "log2" if fused else "raw"
    if _BF16 and q_mode == "log2":
        # the fused window kernels keep the scores in the log2 domain: q is rounded as q * hd^-1/2 * log2(e)
        q = _rb(qkv[0] * (hd ** -0.5 * LOG2E)) / LOG2E
    elif _BF16 and q_mode == "raw":
        # the GEMM route (head_dim 20, C = 768, small windows) rounds the packed qkv rows as the GEMM wrote them and
        # applies the scale to the fp32 scores
        q = _rb(qkv[0]) * hd ** -0.5
    else:
        q = qkv[0] * hd ** -0.5
    k, v = _rb(qkv[1]), _rb(qkv[2])
    att = q @ k.transpose(-2, -1)                                          # [B_, nH, n, n]
    bias = sd[pfx + ".relative_position_bias_table"][relative_position_index(w).reshape(-1)]
    att = att + bias.reshape(n, n, heads).permute(2, 0, 1)[None]
    if mask is not None:
        nW = mask.shape[0]
        att = (att.reshape(B_ // nW, nW, heads, n, n) + mask[None, :, None]).reshape(B_, heads, n, n)
    if _BF16:   # the kernels round the UNNORMALISED numerators and divide the product by the fp32 row sum
        e = (att - att.amax(dim=-1, keepdim=True)).exp()
        out = (_rb(e) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        att = att.softmax(dim=-1)
        out = att @ v
    out = out.transpose(1, 2).reshape(B_, n, C)
    return _lin_mm(out, sd, pfx + ".proj")


def swin_block(x, emb, sd, pfx, res, heads, window, shift):
    """x [B, L, C]; swin_gnn.py:232-277 (window gather/scatter :28-57, roll :247-270)."""
    B, L, C = x.shape
    w, s = eff_window(res, window, shift)
    x = _cond(x, emb, sd, pfx + ".affine")
    short = x
    y = _ln(x, sd, pfx + ".norm1").reshape(B, res, res, C)
    if s > 0:
        y = torch.roll(y, shifts=(-s, -s), dims=(1, 2))
    nw = res // w
    yw = y.reshape(B, nw, w, nw, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nw * nw, w * w, C)
    aw = window_attention(yw, sd, pfx + ".attn", heads, w, shift_mask(res, w, s) if s > 0 else None)
    y = aw.reshape(B, nw, nw, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C)
    if s > 0:
        y = torch.roll(y, shifts=(s, s), dims=(1, 2))
    x = short + y.reshape(B, L, C)
    h = F.gelu(_lin_mm(_ln(x, sd, pfx + ".norm2"), sd, pfx + ".mlp.fc1"))
    return x + _lin_mm(h, sd, pfx + ".mlp.fc2")


def patch_merging(x, sd, pfx, res):
    """[B, res*res, C] -> [B, res*res/4, 2C]; swin_gnn.py:314-335 (channel order :325-329)."""
    B, L, C = x.shape
    g = x.reshape(B, res // 2, 2, res // 2, 2, C)          # [B, Y, dy, X, dx, C]
    g = torch.cat([g[:, :, 0, :, 0], g[:, :, 1, :, 0], g[:, :, 0, :, 1], g[:, :, 1, :, 1]], dim=-1)
    g = g.reshape(B, L // 4, 4 * C)
    return _lin_mm(_ln(g, sd, pfx + ".norm"), sd, pfx + ".reduction")


def patch_breakup(x, sd, pfx, res):
    """[B, res*res, 4C'] -> [B, 4*res*res, C']; swin_gnn.py:374-403 (scatter order :394-397)."""
    B, L, D = x.shape
    Co = D // 4
    y = _ln(_lin_mm(x, sd, pfx + ".pre_linear"), sd, pfx + ".norm").reshape(B, res, res, 4, Co)
    out = torch.empty(B, res, 2, res, 2, Co, dtype=x.dtype)
    for k, (dy, dx) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        out[:, :, dy, :, dx] = y[:, :, :, k]
    out = _ln(out.reshape(B, 4 * L, Co), sd, pfx + ".post_norm")
    return _lin_mm(out, sd, pfx + ".post_linear")


def patch_embed(img, emb, sd, pfx, p):
    """[B, in, N, N] -> [B, (N/p)^2, C]; swin_gnn.py:562-577."""
    y = F.conv2d(img, sd[pfx + ".proj.weight"], sd[pfx + ".proj.bias"], stride=p)
    y = _ln(y.flatten(2).transpose(1, 2), sd, pfx + ".norm")
    return _cond(y, emb, sd, pfx + ".affine")


def read_out(tok, sd, res, p):
    """[B, L, C] -> [B, N, N]; swin_gnn.py:697-701, 750-751."""
    B, L, C = tok.shape
    y = tok.reshape(B, res, res, C).permute(0, 3, 1, 2)
    y = F.conv_transpose2d(y, sd["read_out.0.weight"], sd["read_out.0.bias"], stride=p)
    y = F.conv2d(y, sd["read_out.1.weight"], sd["read_out.1.bias"])
    y = F.conv2d(y, sd["read_out.2.weight"], sd["read_out.2.bias"])
    return y.squeeze(1)


def collapsed_read_out(sd, p):
    """The three read-out convolutions have no non-linearity between them, so they are one
    [C] -> [p*p] affine map per token.  Returns (Weff [C, p*p], beff scalar),
    computed in float64."""
    W0 = sd["read_out.0.weight"].double()            # [Cin, Cout, p, p]
    W1 = sd["read_out.1.weight"].double()[:, :, 0, 0]  # [C1, C0]
    W2 = sd["read_out.2.weight"].double()[0, :, 0, 0]  # [C1]
    v = W2 @ W1                                      # [C0]
    Weff = torch.einsum("o,ioyx->iyx", v, W0).reshape(W0.shape[0], p * p)
    beff = v @ sd["read_out.0.bias"].double() + W2 @ sd["read_out.1.bias"].double() + sd["read_out.2.bias"].double()[0]
    return Weff, beff


def forward_features(img, emb, sd, cfg, trace=None):
    """swin_gnn.py:729-753.  ``trace`` (optional dict) collects intermediate activations."""
    p = cfg["patch_size"]
    x = patch_embed(img, emb, sd, "patch_embed", p)
    if trace is not None:
        trace["patch_embed"] = x
    skips = []
    for st in stages(cfg):
        if not st["up"] and st["prefix"].startswith("up_layers"):
            skips.pop()                                      # first decoder layer drops its skip (:745)
        if st["up"]:
            x = torch.cat([x, skips.pop()], dim=-1)          # :743
            x = patch_breakup(x, sd, st["prefix"] + ".upsample", st["res"] // 2)
        for j in range(st["depth"]):
            shift = 0 if j % 2 == 0 else cfg["window_size"] // 2   # :459
            x = swin_block(x, emb, sd, f"{st['prefix']}.blocks.{j}", st["res"], st["heads"],
                           cfg["window_size"], shift)
        if st["down"]:
            x = patch_merging(x, sd, st["prefix"] + ".downsample", st["res"])
        if st["prefix"].startswith("down_layers"):
            skips.append(x)
        if trace is not None:
            trace[st["prefix"]] = x
    x = _ln(x, sd, "norm")
    return read_out(x, sd, cfg["img_size"] // p, p)


def swingnn_forward(sd, cfg, x, node_flags, noise_labels, self_cond=None, trace=None):
    """SwinGNN.forward, swin_gnn.py:755-779.  x [B,N,N] -> [B,N,N]."""
    emb = noise_embedding(noise_labels, sd, cfg["embed_dim"])
    img = x[:, None]
    if cfg["self_condition"]:
        sc = torch.zeros_like(img) if self_cond is None else self_cond[:, None]
        img = torch.cat([sc, img], dim=1)
    y = forward_features(img, emb, sd, cfg, trace)
    y = mask_adjs(y, node_flags)
    return 0.5 * (y + y.transpose(-1, -2))


# --------------------------------------------------------------------------------------
# EDM preconditioner (model/precond/precond.py, runner/objectives/edm.py)
# --------------------------------------------------------------------------------------
def edm_coefficients(sigmas):
    """runner/objectives/edm.py:122-126."""
    sd2 = SIGMA_DATA ** 2
    c_skip = sd2 / (sigmas ** 2 + sd2)
    c_out = sigmas * SIGMA_DATA / (sigmas ** 2 + sd2).sqrt()
    c_in = 1 / (sd2 + sigmas ** 2).sqrt()
    c_noise = sigmas.log() / 4
    return c_skip, c_out, c_in, c_noise


def _strip(sd, prefix="model."):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def precond_forward(sd, cfg, x, node_flags, sigmas, coin: bool):
    """Precond.forward, model/precond/precond.py:22-53.  ``sd`` carries the 'model.' prefix.

    ``coin`` is the outcome of ``np.random.rand() < 0.5`` (:40); the caller draws it so that the
    numpy stream is consumed exactly as the reference does.  The passed-in self_cond argument of the
    reference is discarded (:39), hence it is not a parameter here."""
    net = _strip(sd)
    c_skip, c_out, c_in, c_noise = (t.reshape(-1, 1, 1) if i < 3 else t
                                    for i, t in enumerate(edm_coefficients(sigmas)))
    sc = None
    if cfg["self_condition"] and coin:
        with torch.no_grad():  # :41-45 (no_grad + detach_): the self-conditioning pass carries no gradient
            sc = mask_adjs(c_skip * x + c_out * swingnn_forward(net, cfg, c_in * x, node_flags, c_noise, None), node_flags)
    Fx = swingnn_forward(net, cfg, c_in * x, node_flags, c_noise, sc)
    return mask_adjs(c_skip * x + c_out * Fx, node_flags)


# --------------------------------------------------------------------------------------
# EDM stochastic Heun sampler (runner/mcmc_sampler/edm.py, shipped defaults)
# --------------------------------------------------------------------------------------
def edm_sigma_steps(num_steps=256, sigma_min=0.002, sigma_max=80.0, rho=7):
    """float64 sigma grid with trailing 0, cast to float32; mcmc_sampler/edm.py:74,90-92,143-148."""
    i = torch.arange(num_steps, dtype=torch.float64)
    s = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([s, torch.zeros_like(s[:1])]).to(torch.float32)


def gen_init_sample(node_flags, raw=None):
    """Folded-normal symmetric init; runner/mcmc_sampler/__init__.py:28-41 (draw on the CPU generator)."""
    B, N = node_flags.shape
    if raw is None:
        raw = torch.randn((B, N, N))
    a = raw.triu(diagonal=1).abs()
    return mask_adjs(a + a.transpose(-1, -2), node_flags)


def edm_sample(denoiser, node_flags, num_steps=256, self_condition=True, init_adjs=None,
               noise_fn=None, sanity_check_gt_adjs=None,
               S_churn=40.0, S_min=0.05, S_max=50.0, S_noise=1.003):
    """EDMSampler.sample with solver='heun', schedule='linear', scaling='none', alpha=1
    (mcmc_sampler/edm.py:125-232).  ``denoiser(x, node_flags, sigmas, self_cond)`` is a Precond-like
    callable.  ``noise_fn(x)`` returns the raw N(0,1) tensor that the reference gets from
    ``torch.randn_like`` (utils/graph_utils.py:117); default torch.randn_like."""
    noise_fn = noise_fn or torch.randn_like
    t = edm_sigma_steps(num_steps)
    if init_adjs is None:
        init_adjs = gen_init_sample(node_flags)
    x_next = init_adjs * t[0]
    sc = None
    B = node_flags.shape[0]
    for i in range(num_steps):
        t_cur, t_next = t[i], t[i + 1]
        x_cur = x_next
        gamma = min(S_churn / num_steps, np.sqrt(2) - 1) if S_min <= t_cur <= S_max else 0      # :172
        t_hat = torch.as_tensor(t_cur + gamma * t_cur)                                          # :173
        x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).clip(min=0).sqrt() * S_noise * sym_noise_from_raw(noise_fn(x_cur))
        x_hat = mask_adjs(x_hat, node_flags)                                                    # :176
        h = t_next - t_hat
        sig = t_hat.reshape(-1).expand(B)
        den = denoiser(x_hat, node_flags, sig, sc) if sanity_check_gt_adjs is None else sanity_check_gt_adjs
        den = mask_adjs(den, node_flags)
        d_cur = mask_adjs((1 / t_hat) * x_hat - (1 / t_hat) * den, node_flags)                  # :189-191
        x_prime = x_hat + h * d_cur
        t_prime = t_hat + h
        if i == num_steps - 1:
            x_next = x_hat + h * d_cur                                                          # :197
        else:
            if sanity_check_gt_adjs is None:
                sc = den if self_condition else None                                            # :203
                den = denoiser(x_hat, node_flags, sig, sc)                                      # :205 (sic: x_hat, t_hat)
            d_prime = (1 / t_prime) * x_prime - (1 / t_prime) * den                             # :211-212
            x_next = x_hat + h * (0.5 * d_cur + 0.5 * d_prime)                                  # :213
        x_next = mask_adjs(x_next, node_flags)
        sc = den if self_condition else None                                                    # :216
        assert symmetry_residual(x_next) < 1e-2                                                 # :217
    return x_next


# --------------------------------------------------------------------------------------
# training objective and loss (runner/objectives/edm.py, loss/rainbow_loss.py)
# --------------------------------------------------------------------------------------
def edm_training_sigmas_weights(rnd_normal):
    """runner/objectives/edm.py:174-176 with the randn draw factored out."""
    sig = (rnd_normal * 1.2 - 1.2).exp()
    w = (sig ** 2 + SIGMA_DATA ** 2) / (sig * SIGMA_DATA) ** 2
    return sig, w


def edm_noisy_input(clean, node_flags, sigmas, raw_noise):
    """runner/objectives/edm.py:181-186 -> utils/graph_utils.py:122-152."""
    return mask_adjs(clean + sym_noise_from_raw(raw_noise) * sigmas.reshape(-1, 1, 1), node_flags)


def edm_loss(pred, target, node_flags, weights):
    """loss/rainbow_loss.py:84-111, objective 'edm', no reweighting, reduction='none',
    regression_loss_weight=1 -> [B]."""
    sq = mask_adjs((pred - target) ** 2 * weights.reshape(-1, 1, 1), node_flags)
    return sq.sum(dim=[-1, -2]) / node_flags.sum(dim=-1) ** 2


def threshold_adj(x, node_flags):
    """Post-sampling quantisation: clamp(-1,1), then 2-type ddpm->int (x > 0), masked.
    runner/sampler/sampler.py:131-135, utils/attribute_code.py:133-159."""
    return mask_adjs((x.clamp(-1.0, 1.0) > 0).to(torch.float32), node_flags)


# --------------------------------------------------------------------------------------
# node + adjacency (molecule) variants: NodeAdjSwinGNN / NodeAdjPrecond / NodeAdjEDMSampler
# --------------------------------------------------------------------------------------
def mask_nodes(nodes: torch.Tensor, flags: torch.Tensor) -> torch.Tensor:
    """Zero node rows whose flag is 0; out-of-place.  utils/graph_utils.py:41-86 (along_dim=None)."""
    keep = flags.to(torch.bool)
    if nodes.dim() == 3:
        keep = keep[..., None]
    return nodes * keep.to(nodes.dtype)


def _adj4(a):
    return a[:, None] if a.dim() == 3 else a                      # swin_gnn.py:966-969


def _feat3(f):
    return f[:, None] if f.dim() == 2 else f.permute(0, 2, 1)     # swin_gnn.py:971-976 -> [B, C, N]


def nodeadj_forward(sd, cfg, adj, node, node_flags, noise_labels, sc_adj=None, sc_node=None):
    """NodeAdjSwinGNN.forward, model/swin_gnn/swin_gnn.py:960-1021.
    adj [B,N,N] | [B,Ca,N,N], node [B,N] | [B,N,Cn] -> (adj_out same layout, node_out same layout).
    cfg carries out_chans_adj / out_chans_node besides the SwinGNN keys."""
    emb = noise_embedding(noise_labels, sd, cfg["embed_dim"])
    a = _adj4(adj)
    f = _feat3(node.to(torch.float32))
    if cfg["self_condition"]:
        a = torch.cat([torch.zeros_like(a) if sc_adj is None else _adj4(sc_adj), a], dim=1)        # :985-986
        f = torch.cat([torch.zeros_like(f) if sc_node is None else _feat3(sc_node), f], dim=1)     # :987-988
    N = f.shape[-1]
    nm = f[..., None].expand(-1, -1, -1, N)                                                        # :991  [B,C,N,N], value of row i
    nm = mask_adjs(torch.cat([nm, nm.transpose(-1, -2)], dim=1), node_flags)                       # :992-994
    img = torch.cat([a, nm], dim=1)                                                                # :996
    p = cfg["patch_size"]
    x = patch_embed(img, emb, sd, "patch_embed", p)
    skips = []
    for st in stages(cfg):
        if not st["up"] and st["prefix"].startswith("up_layers"):
            skips.pop()
        if st["up"]:
            x = patch_breakup(torch.cat([x, skips.pop()], dim=-1), sd, st["prefix"] + ".upsample", st["res"] // 2)
        for j in range(st["depth"]):
            shift = 0 if j % 2 == 0 else cfg["window_size"] // 2
            x = swin_block(x, emb, sd, f"{st['prefix']}.blocks.{j}", st["res"], st["heads"], cfg["window_size"], shift)
        if st["down"]:
            x = patch_merging(x, sd, st["prefix"] + ".downsample", st["res"])
        if st["prefix"].startswith("down_layers"):
            skips.append(x)
    x = _ln(x, sd, "norm")
    B, L, C = x.shape
    res = cfg["img_size"] // p
    y = x.reshape(B, res, res, C).permute(0, 3, 1, 2)
    y = F.conv_transpose2d(y, sd["read_out.0.weight"], sd["read_out.0.bias"], stride=p)
    y = F.conv2d(y, sd["read_out.1.weight"], sd["read_out.1.bias"])
    rep = F.conv2d(y, sd["read_out.2.weight"], sd["read_out.2.bias"])                              # [B, C, N, N]  :955-956

    def mlp(t, pfx):
        return _lin(F.gelu(_lin(t, sd, pfx + ".fc1")), sd, pfx + ".fc2")

    ao = mlp(rep.permute(0, 2, 3, 1), "readout_adj_mlp")                                           # :1000-1003
    ao = ao.squeeze(-1) if cfg["out_chans_adj"] == 1 else ao.permute(0, 3, 1, 2)
    pooled = mask_adjs(rep, node_flags).mean(dim=-1)                                               # :1006-1007 [B, C, N]
    no = mlp(pooled.permute(0, 2, 1), "readout_node_mlp")                                          # :1009-1012
    no = no.squeeze(-1) if cfg["out_chans_node"] == 1 else no
    no = mask_nodes(no, node_flags)
    ao = mask_adjs(ao, node_flags)
    return 0.5 * (ao + ao.transpose(-1, -2)), no                                                   # :1015-1018


def nodeadj_precond_forward(sd, cfg, adjs, nodes, node_flags, sigmas, coin: bool, sc_adjs=None, sc_nodes=None):
    """NodeAdjPrecond.forward, model/precond/precond.py:65-110.  Unlike Precond.forward the passed-in
    self-conditioning tensors ARE used (:81-82) unless the coin replaces them (:90-98)."""
    net = _strip(sd)
    c_skip, c_out, c_in, c_noise = edm_coefficients(sigmas)
    ca = (lambda c: c.reshape(-1, 1, 1, 1)) if adjs.dim() == 4 else (lambda c: c.reshape(-1, 1, 1))
    cf = (lambda c: c.reshape(-1, 1)) if nodes.dim() == 2 else (lambda c: c.reshape(-1, 1, 1))
    if cfg["self_condition"] and coin:
        with torch.no_grad():
            sa, sn = nodeadj_forward(net, cfg, ca(c_in) * adjs, cf(c_in) * nodes, node_flags, c_noise, sc_adjs, sc_nodes)
            sc_adjs = mask_adjs(ca(c_skip) * adjs + ca(c_out) * sa, node_flags)
            sc_nodes = mask_nodes(cf(c_skip) * nodes + cf(c_out) * sn, node_flags)
    Fa, Fn = nodeadj_forward(net, cfg, ca(c_in) * adjs, cf(c_in) * nodes, node_flags, c_noise, sc_adjs, sc_nodes)
    return (mask_adjs(ca(c_skip) * adjs + ca(c_out) * Fa, node_flags),
            mask_nodes(cf(c_skip) * nodes + cf(c_out) * Fn, node_flags))


def nodeadj_gen_init_sample(node_flags, num_node_chan, num_edge_chan, raw_adj=None, raw_node=None):
    """NodeAdjEDMSampler.gen_init_sample (symmetric noise, folded_norm=False as sample() calls it),
    runner/mcmc_sampler/edm.py:261-293."""
    B, N = node_flags.shape
    if raw_adj is None:
        raw_adj = torch.randn((B, num_edge_chan, N, N))
    if raw_node is None:
        raw_node = torch.randn((B, N, num_node_chan))
    a = raw_adj.triu(diagonal=1)
    a = mask_adjs(a + a.transpose(-1, -2), node_flags)
    n = mask_nodes(raw_node, node_flags)
    return (a.squeeze(1) if num_edge_chan == 1 else a), (n.squeeze(-1) if num_node_chan == 1 else n)


def nodeadj_edm_sample(denoiser, node_flags, init_adjs, init_nodes, num_steps=256, self_condition=True,
                       noise_fn=None, sanity_gt=None, S_churn=40.0, S_min=0.05, S_max=50.0, S_noise=1.003):
    """NodeAdjEDMSampler.sample with the shipped defaults (heun, linear schedule, no scaling, alpha=1, symmetric
    noise); runner/mcmc_sampler/edm.py:295-449.  ``denoiser(adjs, nodes, flags, sigmas, sc_adjs, sc_nodes)``;
    ``noise_fn(x)`` supplies the raw normals of randn_like, called for the adjacency first and then the nodes
    (:362-368)."""
    noise_fn = noise_fn or torch.randn_like
    t = edm_sigma_steps(num_steps)
    a_next, n_next = init_adjs * t[0], init_nodes * t[0]
    sca = scn = None
    B = node_flags.shape[0]
    for i in range(num_steps):
        t_cur, t_next = t[i], t[i + 1]
        gamma = min(S_churn / num_steps, np.sqrt(2) - 1) if S_min <= t_cur <= S_max else 0         # :359
        t_hat = torch.as_tensor(t_cur + gamma * t_cur)
        k = (t_hat ** 2 - t_cur ** 2).clip(min=0).sqrt() * S_noise
        a_hat = mask_adjs(a_next + k * sym_noise_from_raw(noise_fn(a_next)), node_flags)           # :362-363, :369
        n_hat = mask_nodes(n_next + k * noise_fn(n_next), node_flags)                              # :367-370
        h = t_next - t_hat
        sig = t_hat.reshape(-1).expand(B)
        if sanity_gt is None:
            da, dn = denoiser(a_hat, n_hat, node_flags, sig, sca, scn)                             # :378
        else:
            da, dn = sanity_gt
        da, dn = mask_adjs(da, node_flags), mask_nodes(dn, node_flags)
        dca = mask_adjs((1 / t_hat) * a_hat - (1 / t_hat) * da, node_flags)                        # :388-391
        dcn = mask_nodes((1 / t_hat) * n_hat - (1 / t_hat) * dn, node_flags)
        a_pr, n_pr = a_hat + h * dca, n_hat + h * dcn
        t_prime = t_hat + h
        if i == num_steps - 1:
            a_next, n_next = a_hat + h * dca, n_hat + h * dcn                                      # :398-400
        else:
            if sanity_gt is None:
                scn = dn if self_condition else None                                               # :406-407
                sca = da if self_condition else None
                da, dn = denoiser(a_hat, n_hat, node_flags, sig, sca, scn)                         # :409 (sic: *_hat, t_hat)
            da, dn = mask_adjs(da, node_flags), mask_nodes(dn, node_flags)
            dpn = (1 / t_prime) * n_pr - (1 / t_prime) * dn                                        # :418-421
            dpa = (1 / t_prime) * a_pr - (1 / t_prime) * da
            n_next = n_hat + h * (0.5 * dcn + 0.5 * dpn)                                           # :422-423
            a_next = a_hat + h * (0.5 * dca + 0.5 * dpa)
        a_next, n_next = mask_adjs(a_next, node_flags), mask_nodes(n_next, node_flags)             # :425-426
        sca = da if self_condition else None                                                       # :427-428
        scn = dn if self_condition else None
        assert symmetry_residual(a_next) < 1e-2                                                    # :429-430
    return a_next, n_next
