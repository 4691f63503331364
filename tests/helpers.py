"""Shared builders for the parity tests: regenerate the weights / inputs of a golden case."""
import numpy as np
import torch

from oracle import filler
from oracle import swingnn_ref as O


def block_state(C, heads, w, seed, prefix="blk", regime="stress"):
    """State dict of one SwinTransformerBlock with the reference's key names and shapes."""
    sd = {
        "affine.weight": (2 * C, 512), "affine.bias": (2 * C,),
        "norm1.weight": (C,), "norm1.bias": (C,),
        "attn.relative_position_bias_table": ((2 * w - 1) ** 2, heads),
        "attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,),
        "attn.proj.weight": (C, C), "attn.proj.bias": (C,),
        "norm2.weight": (C,), "norm2.bias": (C,),
        "mlp.fc1.weight": (4 * C, C), "mlp.fc1.bias": (4 * C,),
        "mlp.fc2.weight": (C, 4 * C), "mlp.fc2.bias": (C,),
    }
    # buffers are part of the reference state dict; the filler leaves them untouched, and key order
    # matters for nothing because the filler hashes by key name
    zeros = {k: torch.zeros(s) for k, s in sd.items()}
    zeros["attn.relative_position_index"] = O.relative_position_index(w)
    filled = filler.fill_state_dict(zeros, seed, regime)
    return {f"{prefix}.{k}": v for k, v in filled.items()}


def attn_state(C, heads, w, seed, prefix="attn"):
    shapes = {
        "relative_position_bias_table": ((2 * w - 1) ** 2, heads),
        "qkv.weight": (3 * C, C), "qkv.bias": (3 * C,), "proj.weight": (C, C), "proj.bias": (C,),
    }
    zeros = {k: torch.zeros(s) for k, s in shapes.items()}
    zeros["relative_position_index"] = O.relative_position_index(w)
    return {f"{prefix}.{k}": v for k, v in filler.fill_state_dict(zeros, seed, "stress").items()}


def merge_state(C, seed, prefix="m"):
    zeros = {"reduction.weight": torch.zeros(2 * C, 4 * C), "norm.weight": torch.zeros(4 * C), "norm.bias": torch.zeros(4 * C)}
    return {f"{prefix}.{k}": v for k, v in filler.fill_state_dict(zeros, seed, "stress").items()}


def breakup_state(dim, seed, prefix="u"):
    Co = dim // 4
    zeros = {"pre_linear.weight": torch.zeros(dim, dim), "norm.weight": torch.zeros(dim), "norm.bias": torch.zeros(dim),
             "post_linear.weight": torch.zeros(Co, Co), "post_norm.weight": torch.zeros(Co), "post_norm.bias": torch.zeros(Co)}
    return {f"{prefix}.{k}": v for k, v in filler.fill_state_dict(zeros, seed, "stress").items()}


def embed_state(C, cin, p, seed, prefix="pe"):
    zeros = {"affine.weight": torch.zeros(2 * C, 512), "affine.bias": torch.zeros(2 * C),
             "proj.weight": torch.zeros(C, cin, p, p), "proj.bias": torch.zeros(C),
             "norm.weight": torch.zeros(C), "norm.bias": torch.zeros(C)}
    return {f"{prefix}.{k}": v for k, v in filler.fill_state_dict(zeros, seed, "stress").items()}


def net_state(meta_keys, seed, regime):
    """Full Precond(SwinGNN) state dict from the golden key/shape table (prefix 'model.')."""
    dt = {"torch.float32": torch.float32, "torch.int64": torch.int64}
    zeros = {}
    for k, (shape, dtype) in meta_keys.items():
        zeros[k] = torch.zeros(shape, dtype=dt[dtype])
    # integer / mask buffers are recomputed by whoever consumes the dict; fill only floats
    return filler.fill_state_dict(zeros, seed, regime)


def net_inputs(case, si):
    cfg = case["cfg"]
    B, N, seed = case["B"], cfg["img_size"], case["seed"]
    sigma = case["sigmas"][si]
    flags = filler.node_flags("flags", B, N, case["n_lo"], case["n_hi"], seed)
    x = filler.sym_adj("x", B, N, flags, seed + si, scale=float(np.sqrt(sigma ** 2 + 0.25)))
    sig = torch.full((B,), float(sigma)) * (1.0 + 0.1 * filler.uniform("sig", (B,), seed)).abs()
    sc = filler.sym_adj("sc", B, N, flags, seed + 100 + si, scale=0.5)
    return flags, x, sig, sc


def max_abs(a, b):
    return float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max())


def oracle_grads(sd, cfg, x, flags, sig, w, clean, coin):
    """Loss [B] and every parameter gradient of mean(edm_loss(precond_forward(...))) by autograd through the oracle."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    full = dict(sd)
    full.update(leaves)
    with torch.enable_grad():
        D = O.precond_forward(full, cfg, x, flags, sig, coin)
        loss = O.edm_loss(D, clean, flags, w)
        loss.mean().backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items() if v.grad is not None}


def train_case_inputs(case, si):
    """Inputs of a TRAIN_CASES entry: flags, noisy x, sigmas, loss weights, clean target."""
    flags, x, sig, _ = net_inputs(case, si)
    clean = O.mask_adjs(torch.sign(x), flags)
    w = (sig ** 2 + 0.25) / (sig * 0.5) ** 2
    return flags, x, sig, w, clean


def nodeadj_inputs(cfg, B, flags, seed, sigma):
    """Inputs of a NODEADJ_CASES entry: adjacency [B,(Ca,)N,N], node features [B,N(,Cn)], self-conditioning pair,
    sigmas [B].  Multi-channel tensors stack independently seeded single-channel draws."""
    N, Ca, Cn = cfg["img_size"], cfg["out_chans_adj"], cfg["out_chans_node"]
    scale = float(np.sqrt(sigma ** 2 + 0.25))

    def adj(name, sc):
        chans = [filler.sym_adj(f"{name}{c}", B, N, flags, seed, scale=sc) for c in range(Ca)]
        return chans[0] if Ca == 1 else torch.stack(chans, dim=1)

    def node(name, sc):
        f = filler.normalish(name, (B, N, Cn), seed) * sc * flags[:, :, None]
        return f[..., 0].contiguous() if Cn == 1 else f

    sig = torch.full((B,), float(max(sigma, 1e-3))) * (1.0 + 0.1 * filler.uniform("sig", (B,), seed)).abs()
    return adj("adj", scale), node("node", scale), adj("sca", 0.5), node("scn", 0.5), sig


def nodeadj_train_inputs(case, si):
    """Inputs of a NODEADJ_TRAIN_CASES entry: flags, noisy adjacency / nodes, sigmas, loss weights, clean targets."""
    cfg = case["cfg"]
    B, N, seed = case["B"], cfg["img_size"], case["seed"]
    flags = filler.node_flags("flags", B, N, case["n_lo"], case["n_hi"], seed)
    adj, node, _, _, sig = nodeadj_inputs(cfg, B, flags, seed + si, case["sigmas"][si])
    clean_a = O.mask_adjs(torch.sign(adj), flags)
    clean_x = O.mask_nodes(torch.sign(node), flags)
    w = (sig ** 2 + 0.25) / (sig * 0.5) ** 2
    return flags, adj, node, sig, w, clean_a, clean_x


def nodeadj_loss(Da, Dn, ta, tn, flags, w):
    """NodeAdjRainbowLoss, reduction 'none', unit loss weights (loss/rainbow_loss.py:209-249) -> ([B], [B])."""
    wa = w.reshape([-1] + [1] * (Da.dim() - 1))
    wn = w.reshape([-1] + [1] * (Dn.dim() - 1))
    n = flags.sum(-1)
    sa = O.mask_adjs((Da - ta) ** 2 * wa, flags)
    sn = O.mask_nodes((Dn - tn) ** 2 * wn, flags)
    la = sa.flatten(1).sum(1) / n ** 2 / (Da.shape[1] if Da.dim() == 4 else 1)
    ln = sn.flatten(1).sum(1) / n / (Dn.shape[2] if Dn.dim() == 3 else 1)
    return la, ln


def oracle_grads_nodeadj(sd, cfg, adj, node, flags, sig, w, clean_a, clean_x, coin):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    full = dict(sd)
    full.update(leaves)
    with torch.enable_grad():
        Da, Dn = O.nodeadj_precond_forward(full, cfg, adj, node, flags, sig, coin)
        la, ln = nodeadj_loss(Da, Dn, clean_a, clean_x, flags, w)
        (la.mean() + ln.mean()).backward()
    return la.detach(), ln.detach(), {k: v.grad for k, v in leaves.items() if v.grad is not None}


def synthetic_graphs(name, count, n_max, n_lo, n_hi, p_edge, seed):
    """[count, n_max, n_max] 0/1 symmetric adjacency matrices of random graphs (filler-driven, reproducible)."""
    flags = filler.node_flags(name + "/flags", count, n_max, n_lo, n_hi, seed)
    u = filler.uniform(name + "/u", (count, n_max, n_max), seed)
    a = torch.triu(((u + 1.0) * 0.5 < p_edge).float(), 1)
    a = a + a.transpose(-1, -2)
    return (a * flags[:, :, None] * flags[:, None, :]).numpy()


def mmd_graph_sets(golden):
    """Fixed (reference set, predicted set) pairs of binarised graphs for the MMD statistics."""
    sets = {}
    for case in ("ego60_8", "ego60_256"):
        pred = golden[f"sampler/{case}/thresholded"].astype(np.float32)
        sets[f"ref_sampler/{case}"] = (synthetic_graphs("testset", 6, pred.shape[-1], 4, 18, 0.3, 91), pred)
    sets["synthetic"] = (synthetic_graphs("setA", 8, 20, 12, 20, 0.35, 92), synthetic_graphs("setB", 7, 20, 10, 20, 0.2, 93))
    return sets


def synthetic_graph_list():
    """Variable-size 0/1 adjacency matrices (numpy), some with isolated nodes, for the dataset-tensorisation golden."""
    out = []
    for k, n in enumerate((5, 12, 20, 9, 16)):
        a = synthetic_graphs(f"ds{k}", 1, n, n, n, 0.25, 100 + k)[0]
        if k == 1:
            a[3, :] = 0; a[:, 3] = 0     # an isolated node inside the graph: flagged as padding (dataloader.py:238)
        out.append(a.astype(np.float64))
    return out
