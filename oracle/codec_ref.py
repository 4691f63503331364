"""CPU oracle of the attribute codecs (TEST INFRASTRUCTURE ONLY; nothing under swingnn_amd/ imports it).

Restates utils/attribute_code.py of the reference (file:line cited per function) in plain torch CPU ops.  Parity
status: PINNED -- tests/test_oracle_codec.py checks every function against fixtures produced by the reference's own
``attribute_converter`` (tests/golden/make_golden_codec.py -> golden_codec.npz).
"""
import math

import torch


def mask_nodes(x, flags):
    """utils/graph_utils.py:41-86: zero the entries of padded nodes; x [B,N] or [B,N,C]."""
    f = flags.to(x.dtype)
    return x * (f if x.dim() == 2 else f[:, :, None])


def mask_adjs(x, flags):
    """utils/graph_utils.py:5-38: zero rows and columns of padded nodes; x [B,N,N] or [B,C,N,N]."""
    f = flags.to(x.dtype)
    m = f[:, :, None] * f[:, None, :]
    return x * (m if x.dim() == 3 else m[:, None])


def ddpm_to_int(x, flags, k):
    """attribute_ddpm_to_int, :124-177 (flag_quantization=True).  x [B,N] or [B,N,N]."""
    step = 2.0 / (k - 1)
    out = torch.full_like(x, -1.0)
    for i in range(k):
        c = -1.0 + i * step
        lo = -float("inf") if i == 0 else c - 0.5 * step
        hi = float("inf") if i == k - 1 else c + step * 0.5
        out[torch.logical_and(x > lo, x <= hi)] = i
    return (mask_adjs if x.dim() == 3 else mask_nodes)(out, flags)


def _channels_last(x, flags, ddpm_range):
    adj = x.dim() == 4
    mask = mask_adjs if adj else mask_nodes
    if ddpm_range:
        x = mask((x + 1.0) / 2.0, flags)                       # :196-197, :227-228
    return (x.permute(0, 2, 3, 1) if adj else x), mask


def one_hot_to_int(x, flags, k, ddpm_range=True):
    """attribute_one_hot_to_int, :213-238.  x [B,N,C] or [B,C,N,N] -> int64."""
    v, mask = _channels_last(x, flags, ddpm_range)
    return mask(v.argmax(dim=-1), flags)


def bits_to_int(x, flags, k, ddpm_range=True, clamp_int=False):
    """attribute_bits_to_int, :180-210 with bin2dec :320-329 (most significant bit first)."""
    v, mask = _channels_last(x, flags, ddpm_range)
    C = v.shape[-1]
    w = 2 ** torch.arange(C - 1, -1, -1).to(v.dtype)
    out = torch.sum(w * v, -1)
    if clamp_int:
        out = torch.clamp(out, min=0, max=k - 1)
    return mask(out, flags)


def int_to_ddpm(xi, flags, k):
    """attribute_int_to_ddpm, :241-255."""
    return (mask_adjs if xi.dim() == 3 else mask_nodes)(2 * xi / (k - 1.0) - 1.0, flags)


def int_to_one_hot(xi, flags, k, ddpm_range=True):
    """attribute_int_to_one_hot, :284-305."""
    out = torch.nn.functional.one_hot(xi.long(), num_classes=k).float()
    adj = out.dim() == 4
    if adj:
        out = out.permute(0, 3, 1, 2)
    if ddpm_range:
        out = 2 * out - 1
    return (mask_adjs if adj else mask_nodes)(out, flags)


def int_to_bits(xi, flags, k, ddpm_range=True):
    """attribute_int_to_bits, :258-281 with dec2bin :308-317."""
    nb = int(math.ceil(math.log2(k)))
    w = 2 ** torch.arange(nb - 1, -1, -1)
    out = xi.long().unsqueeze(-1).bitwise_and(w).ne(0).float()
    adj = out.dim() == 4
    if adj:
        out = out.permute(0, 3, 1, 2)
    if ddpm_range:
        out = 2 * out - 1
    return (mask_adjs if adj else mask_nodes)(out, flags)


def decode_samples(x, flags, encoding, k, is_adj):
    """_decode_node / _decode_adj, runner/sampler/sampler_node_adj.py:178-200."""
    x = x.clamp(-1.0, 1.0)
    if encoding in ("bits", "one_hot"):
        x = torch.where(x > 0.0, torch.ones_like(x), -torch.ones_like(x))
        x = (mask_adjs if is_adj else mask_nodes)(x, flags)
    if encoding == "ddpm":
        return ddpm_to_int(x, flags, k)
    if encoding == "one_hot":
        return one_hot_to_int(x, flags, k, True)
    return bits_to_int(x, flags, k, True, clamp_int=True)
