"""Deterministic tensor filler (test infrastructure).

Golden fixtures would be huge if they carried weights (35.9 M parameters at
dim-96), so both sides -- the reference modules when the goldens are made, and
the oracle / HIP path when they are checked -- regenerate every weight and every
input from ``(name, shape, seed)`` with the integer hash below.  Only integer
arithmetic and one int->float64 conversion are involved, so the values are
bit-identical on any machine.
"""
import numpy as np
import torch

_MASK = (1 << 64) - 1


def _fnv1a(name: str, seed: int) -> int:
    h = 0xCBF29CE484222325 ^ (seed & _MASK)
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, seed: int = 0) -> torch.Tensor:
    """float32 tensor of i.i.d. values in [-1, 1), a pure function of (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64(_fnv1a(name, seed))
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
        bits = _splitmix64(ctr)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))  # [0,1)
    return torch.from_numpy((2.0 * u - 1.0).astype(np.float32)).reshape(tuple(shape))


def normalish(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Roughly N(0,1): sum of 4 uniforms, rescaled (Irwin-Hall); bit-reproducible."""
    acc = sum(uniform(f"{name}#{k}", shape, seed) for k in range(4))
    return acc * float(np.sqrt(3.0 / 4.0))


def fill_state_dict(sd: dict, seed: int = 0, regime: str = "stress") -> dict:
    """Return a new state dict with every floating parameter replaced by filler values.

    regime 'stress': weights ~ U(-a, a) with a = sqrt(3 / fan_in)  (unit gain, O(1) activations,
                     non-zero biases, LayerNorm gains != 1) -- exposes indexing / bias / ordering bugs.
    regime 'init'  : magnitudes of the reference's own initialisation (|w| <~ 0.035, like
                     trunc_normal_(std=.02)), but with small non-zero biases.
    Integer / mask buffers (relative_position_index, attn_mask) are kept as they are.
    """
    out = {}
    for key, val in sd.items():
        if (not torch.is_floating_point(val)) or key.endswith("attn_mask"):
            out[key] = val.clone()
            continue
        shape = tuple(val.shape)
        u = uniform(key, shape, seed)
        if key.endswith("relative_position_bias_table"):
            t = u * (0.5 if regime == "stress" else 0.04)
        elif val.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = float(np.sqrt(3.0 / fan_in)) if regime == "stress" else 0.0346
            t = u * a
        elif "norm" in key and key.endswith("weight"):
            t = 1.0 + (0.2 if regime == "stress" else 0.02) * u
        else:  # biases (Linear, conv, LayerNorm)
            t = (0.1 if regime == "stress" else 0.02) * u
        out[key] = t.to(val.dtype)
    return out


def node_flags(name: str, batch: int, n_max: int, n_lo: int, n_hi: int, seed: int = 0) -> torch.Tensor:
    """[B, N] float32 0/1 flags: the first n_b entries are 1, n_b drawn in [n_lo, n_hi]."""
    u = uniform(name, (batch,), seed)
    n = (n_lo + ((u + 1.0) * 0.5 * (n_hi - n_lo + 1)).floor().clamp(max=n_hi - n_lo)).to(torch.int64)
    idx = torch.arange(n_max)[None, :]
    return (idx < n[:, None]).to(torch.float32)


def sym_adj(name: str, batch: int, n_max: int, flags: torch.Tensor, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    """[B, N, N] symmetric float32 matrix with zero diagonal, zeroed outside the flagged block."""
    u = normalish(name, (batch, n_max, n_max), seed) * scale
    a = torch.triu(u, 1)
    a = a + a.transpose(-1, -2)
    return a * flags[:, :, None] * flags[:, None, :]
