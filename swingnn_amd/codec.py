"""Attribute codecs of the molecule path on the device (utils/attribute_code.py of the reference):
``attribute_converter`` with the reference's signature over swg_attr_decode / swg_attr_encode, and ``decode_samples``,
the post-sampling quantisation of runner/sampler/sampler_node_adj.py:178-200 in one launch.

Tensors live on the MI355X; integer attributes are float32 (exact), one-hot -> int returns int64 like torch.argmax.
The reference's ``attribute_sanity_check`` asserts (host syncs over ``unique()``) are not run.
"""
import math

import torch

from . import _lib as L
from .engine import _as_f32, stream_ptr

_ENC = {"ddpm": 0, "one_hot": 1, "bits": 2}


def _intervals(num_type, device):
    """Interval bounds of attribute_ddpm_to_int (:133-150), computed in double like the reference, compared in fp32."""
    assert num_type >= 2
    step = 2.0 / (num_type - 1)
    lo, hi = [], []
    for i in range(num_type):
        c = -1.0 + i * step
        lo.append(-float("inf") if i == 0 else c - 0.5 * step)
        hi.append(float("inf") if i == num_type - 1 else c + step * 0.5)
    return (torch.tensor(lo, dtype=torch.float32, device=device), torch.tensor(hi, dtype=torch.float32, device=device))


def _layout(t, encoding, flag_adjs):
    """(B, N, C, is_adj) of an attribute tensor in `encoding`."""
    multi = encoding in ("one_hot", "bits")
    if flag_adjs:
        if t.dim() != (4 if multi else 3):
            raise ValueError(f"adjacency attribute in {encoding!r} must be {'[B,C,N,N]' if multi else '[B,N,N]'}")
        return t.shape[0], t.shape[-1], (t.shape[1] if multi else 1), 1
    if t.dim() != (3 if multi else 2):
        raise ValueError(f"node attribute in {encoding!r} must be {'[B,N,C]' if multi else '[B,N]'}")
    return t.shape[0], t.shape[1], (t.shape[2] if multi else 1), 0


def _decode(x, flags, encoding, num_attr_type, flag_adjs, ddpm_range, binarise, clamp_int):
    x = _as_f32(x)
    if x.device.type != "cuda":
        raise RuntimeError("swingnn_amd.codec runs on an MI355X; there is no CPU fallback")
    B, N, C, adj = _layout(x, encoding, flag_adjs)
    lo = hi = None
    if encoding == "ddpm":
        lo, hi = _intervals(num_attr_type, x.device)
    out = torch.empty((B, N, N) if adj else (B, N), dtype=torch.float32, device=x.device)
    L.check(L.lib.swg_attr_decode(_ENC[encoding], B, N, C, num_attr_type, adj, int(ddpm_range), int(binarise),
                                  int(clamp_int), x.data_ptr(), flags.data_ptr(),
                                  None if lo is None else lo.data_ptr(), None if hi is None else hi.data_ptr(),
                                  out.data_ptr(), stream_ptr()), "swg_attr_decode")
    return out.long() if encoding == "one_hot" else out


def _encode(xi, flags, encoding, num_attr_type, flag_adjs, ddpm_range):
    xi = _as_f32(xi)
    B, N = xi.shape[0], xi.shape[-1]
    adj = int(flag_adjs)
    C = {"ddpm": 1, "one_hot": num_attr_type, "bits": int(math.ceil(math.log2(num_attr_type)))}[encoding]
    if encoding == "ddpm":
        shape = xi.shape
    else:
        shape = (B, C, N, N) if adj else (B, N, C)
    out = torch.empty(shape, dtype=torch.float32, device=xi.device)
    L.check(L.lib.swg_attr_encode(_ENC[encoding], B, N, C, num_attr_type, adj, int(ddpm_range), xi.data_ptr(),
                                  flags.data_ptr(), out.data_ptr(), stream_ptr()), "swg_attr_encode")
    return out


def attribute_converter(in_attr, attr_flags, in_encoding, out_encoding, num_attr_type, flag_nodes=False, flag_adjs=False,
                        flag_in_ddpm_range=True, flag_out_ddpm_range=True, flag_clamp_int=False):
    """utils/attribute_code.py:14-62 (same arguments).  attr_flags: node flags [B,N]."""
    assert (bool(flag_adjs) + bool(flag_nodes)) == 1, "flag_nodes and flag_adjs cannot be both True or False"
    for e in (in_encoding, out_encoding):
        if e not in ("int", "ddpm", "bits", "one_hot"):
            raise ValueError("encoding should be 'int', 'ddpm', 'bits' or 'one_hot'")
    flags = _as_f32(attr_flags.to(in_attr.device))
    if flags.dim() != 2:
        raise NotImplementedError("swingnn_amd.codec: attr_flags are node flags [B, N]")
    # the reference infers adjacency / node from the tensor rank for the scalar encodings (get_mask_func, :65-86)
    if in_encoding == "int":
        int_attr = in_attr
    else:
        adj_in = in_attr.dim() == (4 if in_encoding in ("bits", "one_hot") else 3)
        int_attr = _decode(in_attr, flags, in_encoding, num_attr_type, adj_in, flag_in_ddpm_range, False, flag_clamp_int)
    if out_encoding == "int":
        return int_attr
    return _encode(int_attr, flags, out_encoding, num_attr_type, int_attr.dim() == 3, flag_out_ddpm_range)


def decode_samples(samples, node_flags, encoding_method, num_attr_type, is_adj):
    """_decode_node / _decode_adj of runner/sampler/sampler_node_adj.py:178-200: clamp to [-1,1]; for bits / one-hot the
    channels are binarised at 0 and masked; then attribute_converter(..., out_encoding='int', flag_clamp_int = bits)."""
    flags = _as_f32(node_flags.to(samples.device))
    return _decode(samples, flags, encoding_method, num_attr_type, bool(is_adj), True, True, encoding_method == "bits")
