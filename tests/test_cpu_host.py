"""CPU-side checks (no GPU): the C-ABI library loads and exports everything include/swingnn_hip.h declares,
host-only entry points behave, the module tree matches the reference's state_dict contract and seeded
initialisation, and the host-prepared weight images / step constants are what the kernels assume."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import swingnn_ref as O
from tests.golden.cases import NET_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def S():
    import swingnn_amd
    return swingnn_amd


def build(S, cfg):
    net = S.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                    window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=cfg["self_condition"])
    return S.Precond("edm", net, cfg["self_condition"])


def test_every_declared_symbol_is_exported_and_bound(S):
    from swingnn_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "swingnn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(swg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    raw = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in the header but not exported by the library"
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    assert L.lib.swg_version() >= 1
    names = {L.lib.swg_prof_name(k).decode() for k in range(L.lib.swg_prof_kernels())}
    assert {"k_gemm_nt<32>", "k_gemm_nt_big", "k_wattn_ring<384>", "k_mlp_stream<384>"} <= names and "" not in names


def test_host_only_error_paths(S):
    """Argument validation happens before any HIP call, so it can be exercised without a GPU."""
    from swingnn_amd import _lib as L
    assert L.lib.swg_heun_churn(0, 8, None, None, None, 0.0, None, None) == -4      # SWG_E_NULL
    assert L.lib.swg_mask_adjs(0, 8, 1, 1, None) == -1                               # SWG_E_SHAPE (B = 0)
    assert L.lib.swg_wattn_fwd(None, 0, None, None, 0, None, None, 0, None) == -1
    assert L.lib.swg_wattn_fwd(None, 1, None, None, 0, None, None, 0, None) == -4
    assert L.lib.swg_workspace_bytes(None, 4) == 0
    with pytest.raises(L.SwgError):
        L.check(-2, "x")


@pytest.mark.parametrize("name", list(NET_CASES))
def test_state_dict_contract(S, golden_meta, name):
    pre = build(S, NET_CASES[name]["cfg"])
    sd = pre.state_dict()
    ref = golden_meta[f"keys/{name}"]
    assert set(sd) == set(ref)
    for k, v in sd.items():
        assert list(v.shape) == ref[k][0] and str(v.dtype) == ref[k][1], k
    assert sum(p.numel() for p in pre.parameters()) == golden_meta[f"nparams/{name}"]
    # registered buffers carry the reference's values (strict loading of released checkpoints)
    for k, v in sd.items():
        if k.endswith("relative_position_index"):
            w = int(round(v.shape[0] ** 0.5))
            assert torch.equal(v, O.relative_position_index(w))
        if k.endswith("attn_mask"):
            blk = pre.get_submodule(k.rsplit(".", 1)[0])
            assert torch.equal(v, O.shift_mask(blk.input_resolution[0], blk.window_size, blk.shift_size))


@pytest.mark.parametrize("name", ["ego60", "grid192"])
def test_seeded_init_matches_reference(S, golden, golden_meta, name):
    """Same module creation order and init calls as the reference => identical weights under the same seed."""
    torch.manual_seed(1234)
    sd = build(S, NET_CASES[name]["cfg"]).state_dict()
    keys = golden_meta[f"init_keys/{name}"]
    sums = np.array([float(sd[k].double().sum()) for k in keys])
    asums = np.array([float(sd[k].double().abs().sum()) for k in keys])
    assert np.allclose(sums, golden[f"init/{name}/sums"], rtol=0, atol=1e-9)
    assert np.allclose(asums, golden[f"init/{name}/abs_sums"], rtol=1e-12)


def test_window_clamp_and_unsupported_options(S):
    blk = S.SwinTransformerBlock(96, 512, (8, 8), 3, window_size=24, shift_size=12)
    assert (blk.window_size, blk.shift_size) == (8, 0) and blk.attn_mask is None
    blk = S.SwinTransformerBlock(96, 512, (12, 12), 3, window_size=6, shift_size=3)
    assert blk.attn_mask.shape == (4, 36, 36)
    with pytest.raises(NotImplementedError):
        S.SwinGNN(img_size=24, patch_size=3, in_chans=1, embed_dim=60, depths=[1], drop_path_rate=0.1)
    with pytest.raises(NotImplementedError):
        S.Precond("vp", build(S, NET_CASES["ego60"]["cfg"]).model, True)
    with pytest.raises(RuntimeError):  # parameters on the CPU: there is no fallback
        pre = build(S, NET_CASES["ego60"]["cfg"]).eval()
        with torch.no_grad():
            pre(torch.zeros(1, 24, 24), torch.ones(1, 24), torch.ones(1))


def test_sampler_constants_match_oracle(S):
    smp = S.EDMSampler(num_steps=256, clip_samples=False, clip_samples_min=-1., clip_samples_max=1.,
                       clip_samples_scope="x_0", self_condition=True, dev="cpu")
    t = O.edm_sigma_steps(256)
    assert torch.equal(smp.sigma_steps.to(torch.float32), t[:-1])
    consts = smp.step_constants()
    assert len(consts) == 256
    for i in (0, 1, 17, 100, 254, 255):
        t_cur, t_next = t[i], t[i + 1]
        gamma = min(40 / 256, np.sqrt(2) - 1) if 0.05 <= t_cur <= 50 else 0
        t_hat = torch.as_tensor(t_cur + gamma * t_cur)
        coef = (t_hat ** 2 - t_cur ** 2).clip(min=0).sqrt() * 1.003
        assert consts[i] == (float(t_hat), float(coef), float(t_next - t_hat), float(t_hat + (t_next - t_hat)))
    assert consts[0][1] == 0.0 and consts[255][3] == 0.0  # sigma 80 > S_max: no churn; last step lands on t = 0
    # static helpers used by the reference's sampler utilities
    flags = torch.tensor([[1., 1., 0.]])
    a = torch.tensor([[[0.2, 0.7, 0.9], [0.7, 0.1, 0.9], [0.9, 0.9, 0.9]]])
    assert torch.equal(S.EDMSampler.adj_to_int(a, flags, 0.5), torch.tensor([[[0., 1., 0.], [1., 0., 0.], [0., 0., 0.]]]))
    assert float(S.EDMSampler.get_num_edges(a, flags, 0.5)) == 1.0


@pytest.mark.parametrize("Cdim", [96, 192])
def test_mlp_image_is_what_the_kernel_indexes(S, Cdim):
    """Rebuild fc1 / fc2 from the image with the kernel's own address arithmetic (csrc/mlp_fused.hip)."""
    from swingnn_amd import engine as E
    g = torch.Generator().manual_seed(Cdim)
    w1 = torch.randn(4 * Cdim, Cdim, generator=g)
    w2 = torch.randn(Cdim, 4 * Cdim, generator=g)
    img = E.mlp_image(w1, w2).float().numpy()
    w1b, w2b = w1.to(torch.bfloat16).float().numpy(), w2.to(torch.bfloat16).float().numpy()
    HC, KS, NT = 64, Cdim // 32, Cdim // 16
    swz = lambda r, ch: ch ^ ((-(r >> 2)) & 3)
    rng = np.random.default_rng(0)
    for _ in range(400):
        hc, lane = int(rng.integers(0, 4 * Cdim // HC)), int(rng.integers(0, 64))
        l15, lg = lane & 15, lane >> 4
        chunk = img[hc]
        # fc1 A-operand fragment (nt, ks): row nt*16+l15 of the chunk, k = 32 ks + 8 lg + e
        nt, ks = int(rng.integers(0, 4)), int(rng.integers(0, KS))
        r = nt * 16 + l15
        off = ((r * 4 + swz(r, lg)) * 16 + ks * HC * 64) // 2
        assert np.array_equal(chunk[off:off + 8], w1b[hc * HC + r, 32 * ks + 8 * lg: 32 * ks + 8 * lg + 8])
        # fc2 A-operand fragment (nt2, s2): image row -> channel rho; k-slot e -> hidden pi(lg, e)
        nt2, s2 = int(rng.integers(0, NT)), int(rng.integers(0, 2))
        r2 = nt2 * 16 + l15
        off2 = (HC * Cdim * 2 + (r2 * 4 + swz(r2, lg)) * 16 + s2 * Cdim * 64) // 2
        chan = 32 * (nt2 >> 1) + 8 * (l15 >> 2) + 4 * (nt2 & 1) + (l15 & 3)
        hid = [hc * HC + 32 * s2 + (4 * lg + e if e < 4 else 16 + 4 * lg + e - 4) for e in range(8)]
        assert np.array_equal(chunk[off2:off2 + 8], w2b[chan, hid])


def test_collapsed_read_out_weights(S, golden_meta):
    from swingnn_amd import engine as E
    from tests import helpers as H
    pre = build(S, NET_CASES["grid192"]["cfg"])
    sd = H.net_state(golden_meta["keys/grid192"], 3, "stress")
    pre.load_state_dict({k: v for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith("attn_mask")},
                        strict=False)
    w, b = E.collapse_read_out(pre.model.read_out, 4)
    w_ref, b_ref = O.collapsed_read_out(O._strip(sd), 4)
    assert torch.allclose(w.double(), w_ref, atol=1e-7) and abs(b - float(b_ref)) < 1e-6


# ---- molecule path (NodeAdjSwinGNN / NodeAdjPrecond), host side ---------------------------------------------
def build_nodeadj(S, cfg):
    net = S.NodeAdjSwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=cfg["in_chans"],
                           embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                           window_size=cfg["window_size"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                           drop_path_rate=0.0, self_condition=cfg["self_condition"], symmetric_noise=True,
                           out_chans_adj=cfg["out_chans_adj"], out_chans_node=cfg["out_chans_node"])
    return S.NodeAdjPrecond("edm", net, cfg["self_condition"], symmetric_noise=True)


def test_nodeadj_state_dict_contract_and_seeded_init(S, golden_nodeadj):
    from tests.golden.cases import NODEADJ_CASES
    g, meta = golden_nodeadj
    for name, c in NODEADJ_CASES.items():
        pre = build_nodeadj(S, c["cfg"])
        sd = pre.state_dict()
        ref = meta[f"keys/{name}"]
        assert set(sd) == set(ref), name
        for k, v in sd.items():
            assert list(v.shape) == ref[k][0] and str(v.dtype) == ref[k][1], k
        assert sum(p.numel() for p in pre.parameters()) == meta[f"nparams/{name}"]
    torch.manual_seed(1234)
    sd = build_nodeadj(S, NODEADJ_CASES["qm9_onehot96"]["cfg"]).state_dict()
    keys = meta["init_keys/qm9_onehot96"]
    sums = np.array([float(sd[k].double().sum()) for k in keys])
    asums = np.array([float(sd[k].double().abs().sum()) for k in keys])
    assert np.allclose(sums, g["nainit/qm9_onehot96/sums"], rtol=0, atol=1e-9)
    assert np.allclose(asums, g["nainit/qm9_onehot96/abs_sums"], rtol=1e-12)


def test_nodeadj_collapsed_read_out(S, golden_nodeadj):
    """The three read-out convolutions as one [C] -> [p*p, C] map: equal to running them (float64)."""
    import torch.nn.functional as F
    from swingnn_amd import engine as E
    from tests import helpers as H
    from tests.golden.cases import NODEADJ_CASES
    _, meta = golden_nodeadj
    cfg = NODEADJ_CASES["zinc_ddpm96"]["cfg"]
    pre = build_nodeadj(S, cfg)
    sd = H.net_state(meta["keys/zinc_ddpm96"], 5, "stress")
    pre.load_state_dict({k: v for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith("attn_mask")},
                        strict=False)
    p, C0 = cfg["patch_size"], cfg["embed_dim"]
    w, b = E.collapse_read_out_nodeadj(pre.model.read_out, p)          # [p*p, Cout, Cin], [Cout]
    x = torch.randn(2, C0, 3, 3, dtype=torch.float64)
    ro = pre.model.read_out
    y = F.conv_transpose2d(x, ro[0].weight.double(), ro[0].bias.double(), stride=p)
    y = F.conv2d(F.conv2d(y, ro[1].weight.double(), ro[1].bias.double()), ro[2].weight.double(), ro[2].bias.double())
    mine = torch.einsum("koi,biyx->bykxo", w, x) + b                   # [B, Y, k = i*p+j, X, Cout]
    mine = mine.reshape(2, 3, p, p, 3, C0).permute(0, 5, 1, 2, 4, 3).reshape(2, C0, 3 * p, 3 * p)
    assert torch.allclose(mine, y, atol=1e-9)


def test_workspace_plans_scale_with_batch(S):
    """The host-only planning entry points (swg_workspace_bytes / swg_train_workspace_bytes / the NodeAdj variants) must
    grow monotonically and ~linearly with the batch, also past 2 GiB per term (a max() over byte counts that bound to the
    int overload once dropped every term above 2^31: training grid graphs at B >= 56 failed with SWG_E_WORKSPACE)."""
    from swingnn_amd import engine as E, _lib as L
    from tests.golden.cases import NODEADJ_CASES
    E.HOST_DESCRIPTORS_FOR_TESTS = True
    try:
        grid = dict(img_size=384, patch_size=4, in_chans=1, embed_dim=96, depths=[1, 1, 3, 1], num_heads=[3, 6, 12, 24],
                    window_size=6, self_condition=True)
        nets = [(build(S, grid).model, (8, 48, 64, 112, 256)),
                (build_nodeadj(S, NODEADJ_CASES["qm9_onehot96"]["cfg"]).model, (64, 1280, 10240, 40960))]
        for model, batches in nets:
            cache = E.NetCache(model)
            fwd = [cache.ws_bytes(B) for B in batches]
            if cache.nodeadj is not None:
                trn = [int(L.lib.swg_nodeadj_train_workspace_bytes(C.byref(cache.net), C.byref(cache.nodeadj), B)) for B in batches]
            else:
                trn = [int(L.lib.swg_train_workspace_bytes(C.byref(cache.net), B)) for B in batches]
            for sizes in (fwd, trn):
                assert all(s > 0 for s in sizes)
                for (b0, s0), (b1, s1) in zip(zip(batches, sizes), zip(batches[1:], sizes[1:])):
                    assert s1 > s0, (b0, s0, b1, s1)
                    if s0 > (1 << 30):  # past the fixed split-K buffers the plan is linear in the batch
                        assert abs(s1 / s0 - b1 / b0) < 0.1 * b1 / b0, (b0, s0, b1, s1)
            assert trn[-1] > (1 << 32)
    finally:
        E.HOST_DESCRIPTORS_FOR_TESTS = False
