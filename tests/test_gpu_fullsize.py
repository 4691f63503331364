"""GPU parity at the EXACT bench geometries (BASELINE.json configs[3] grid N=384 and configs[4] DD N=512, dim-96):
`Precond.forward` through the HIP path against the CPU oracle evaluated here (the oracle is pinned to the reference on
the reduced geometries by tests/test_oracle_golden.py; the full-size stages differ only in window counts, e.g. 4
un-clamped windows per graph at C=768 and 16 at the shifted C=384 stage).  Plus block cases at C=768 with un-clamped
windows, and full 256-step trajectories on the grid geometry and on community-small dim-96 against reference goldens
(tests/golden/make_golden_sampler_long.py).

Bars: max-abs error on the denoised adjacency <= 1e-2; thresholded adjacency identical on >= 99.9 % of the valid
entries -- asserted RAW (every valid entry) and filtered (entries further than 1e-2 from the decision boundary).
"""
import os

import numpy as np
import pytest
import torch

from oracle import filler
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import HEADS, NET_CASES, SAMPLER_FULL_CASES, SAMPLER_LONG_CASES
from tests.test_gpu_parity import DEV, load, rel_close, strip

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

FULL_CASES = {
    # config/edm_swin_gnn/grid_edm_swin_gnn_80.yaml with --feature_dims 96: all graphs 361 nodes (bench.py) + a ragged one
    "grid384": dict(img=384, p=4, dim=96, depths=[1, 1, 3, 1], window=6, n=[361, 170], sigmas=[0.3, 6.0], seed=91),
    # config/edm_swin_gnn/dd_edm_swin_gnn_734.yaml with --feature_dims 96
    "dd512": dict(img=512, p=4, dim=96, depths=[1, 1, 3, 1], window=8, n=[500, 233], sigmas=[1.5], seed=92),
}


def agreement(out, ref, flags):
    valid = (flags[:, :, None] * flags[:, None, :]) > 0
    same = (out > 0) == (ref > 0)
    raw = float(same[valid].float().mean())
    sure = valid & (ref.abs() > 1e-2)
    filt = float(same[sure].float().mean()) if sure.any() else 1.0
    return raw, filt


@pytest.mark.parametrize("name", list(FULL_CASES))
def test_precond_full_size(name, monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import swingnn_amd as S
    c = FULL_CASES[name]
    cfg = dict(img_size=c["img"], patch_size=c["p"], in_chans=1, embed_dim=c["dim"], depths=c["depths"],
               num_heads=HEADS, window_size=c["window"], self_condition=True)
    net = S.SwinGNN(img_size=c["img"], patch_size=c["p"], in_chans=1, embed_dim=c["dim"], depths=c["depths"],
                    num_heads=HEADS, window_size=c["window"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=True)
    pre = S.Precond("edm", net, True)
    sd = filler.fill_state_dict(pre.state_dict(), c["seed"], "init")
    pre.load_state_dict(sd)
    pre = pre.to(DEV).eval()
    N = c["img"]
    for bi, n_nodes in enumerate(c["n"]):          # B = 1 per call keeps the oracle at a fraction of a second
        flags = (torch.arange(N)[None, :] < n_nodes).to(torch.float32)
        for si, sigma in enumerate(c["sigmas"]):
            x = filler.sym_adj("x", 1, N, flags, c["seed"] + 10 * bi + si, scale=float(np.sqrt(sigma ** 2 + 0.25)))
            sig = torch.full((1,), float(sigma))
            for coin, u in ((True, 0.25), (False, 0.75)):
                monkeypatch.setattr(np.random, "rand", lambda u=u: u)
                out = pre(x.to(DEV), flags.to(DEV), sig.to(DEV)).cpu()
                monkeypatch.undo()
                ref = O.precond_forward(sd, cfg, x, flags, sig, coin)
                assert float((out - out.transpose(-1, -2)).abs().max()) == 0.0
                err = float((out - ref).abs().max())
                raw, filt = agreement(out, ref, flags)
                print(f"{name} n={n_nodes} sigma={sigma} coin={coin}: max err {err:.3g}, agreement raw {raw:.5f} "
                      f"filtered {filt:.5f}")
                assert err <= 1e-2, err
                assert filt >= 0.999, filt
                assert raw >= 0.999, raw
        pre.assert_symmetric()


def test_precond_grid384_batched_pass_matches_oracle(monkeypatch):
    """The bench geometry at a batch large enough for the batch-dependent kernel choices of the C = 384 stage (32 graphs
    x 576 tokens >= 16384: streamed MLP, fused attention-tail + MLP kernel, the next block's LayerNorm1 produced by the
    previous block's tail), which the B = 1 cases above do not reach.  Every graph is held to the oracle."""
    import swingnn_amd as S
    c = FULL_CASES["grid384"]
    cfg = dict(img_size=c["img"], patch_size=c["p"], in_chans=1, embed_dim=c["dim"], depths=c["depths"],
               num_heads=HEADS, window_size=c["window"], self_condition=True)
    net = S.SwinGNN(img_size=c["img"], patch_size=c["p"], in_chans=1, embed_dim=c["dim"], depths=c["depths"],
                    num_heads=HEADS, window_size=c["window"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=True)
    pre = S.Precond("edm", net, True)
    sd = filler.fill_state_dict(pre.state_dict(), c["seed"], "init")
    pre.load_state_dict(sd)
    pre = pre.to(DEV).eval()
    N, B = c["img"], 32
    n_nodes = torch.tensor([361 - 7 * (i % 5) for i in range(B)])
    flags = (torch.arange(N)[None, :] < n_nodes[:, None]).to(torch.float32)
    sig = torch.tensor([0.05 + 0.4 * i for i in range(B)], dtype=torch.float32)
    x = torch.cat([filler.sym_adj("x", 1, N, flags[i:i + 1], c["seed"] + 100 + i,
                                  scale=float(np.sqrt(float(sig[i]) ** 2 + 0.25))) for i in range(B)])
    monkeypatch.setattr(np.random, "rand", lambda: 0.25)   # self-conditioning branch: two passes
    out = pre(x.to(DEV), flags.to(DEV), sig.to(DEV)).cpu()
    monkeypatch.undo()
    pre.assert_symmetric()
    ref = O.precond_forward(sd, cfg, x, flags, sig, True)
    err = float((out - ref).abs().max())
    raw, filt = agreement(out, ref, flags)
    print(f"grid384 B={B}: max err {err:.3g}, agreement raw {raw:.5f} filtered {filt:.5f}")
    assert err <= 1e-2, err
    assert filt >= 0.999 and raw >= 0.999, (raw, filt)


def test_precond_dd512_batched_pass_matches_oracle(monkeypatch):
    """The DD geometry (N = 512, 8x8 windows) at B = 4: the smallest batch at which the one-pass PatchMerging
    (C = 96: 4 x 64^2 = 16384 merged tokens) and PatchBreakup (Cin = 384: 4 x 64^2 input tokens) kernels run inside the
    network pass; every graph is held to the oracle."""
    import swingnn_amd as S
    c = FULL_CASES["dd512"]
    cfg = dict(img_size=c["img"], patch_size=c["p"], in_chans=1, embed_dim=c["dim"], depths=c["depths"],
               num_heads=HEADS, window_size=c["window"], self_condition=True)
    net = S.SwinGNN(img_size=c["img"], patch_size=c["p"], in_chans=1, embed_dim=c["dim"], depths=c["depths"],
                    num_heads=HEADS, window_size=c["window"], mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                    drop_path_rate=0.0, self_condition=True)
    pre = S.Precond("edm", net, True)
    sd = filler.fill_state_dict(pre.state_dict(), c["seed"], "init")
    pre.load_state_dict(sd)
    pre = pre.to(DEV).eval()
    N, B = c["img"], 4
    n_nodes = torch.tensor([500, 233, 401, 512])
    flags = (torch.arange(N)[None, :] < n_nodes[:, None]).to(torch.float32)
    sig = torch.tensor([0.1, 1.5, 4.0, 20.0], dtype=torch.float32)
    x = torch.cat([filler.sym_adj("x", 1, N, flags[i:i + 1], c["seed"] + 200 + i,
                                  scale=float(np.sqrt(float(sig[i]) ** 2 + 0.25))) for i in range(B)])
    monkeypatch.setattr(np.random, "rand", lambda: 0.25)   # self-conditioning branch: two passes
    out = pre(x.to(DEV), flags.to(DEV), sig.to(DEV)).cpu()
    monkeypatch.undo()
    pre.assert_symmetric()
    ref = O.precond_forward(sd, cfg, x, flags, sig, True)
    err = float((out - ref).abs().max())
    raw, filt = agreement(out, ref, flags)
    print(f"dd512 B={B}: max err {err:.3g}, agreement raw {raw:.5f} filtered {filt:.5f}")
    assert err <= 1e-2, err
    assert filt >= 0.999 and raw >= 0.999, (raw, filt)


# PatchMerging at token counts where the one-pass kernel (csrc/merge_fused.hip) runs: >= 16384 merged tokens, C = 96 /
# 192; the reduced golden geometries stay on the LayerNorm pass + GEMM.  Ragged cases: merged token counts that are
# not a multiple of the 128-token workgroup, odd merged side.
MERGE_FUSED_CASES = {
    "c96_r96_b8": dict(C=96, res=96, B=8, seed=71),
    "c96_r90_b9": dict(C=96, res=90, B=9, seed=72),
    "c192_r48_b32": dict(C=192, res=48, B=32, seed=73),
    "c192_r50_b27": dict(C=192, res=50, B=27, seed=74),
}


@pytest.mark.parametrize("name", list(MERGE_FUSED_CASES))
def test_patch_merging_one_pass_kernel_matches_oracle_and_gemm_path(name):
    import swingnn_amd as S
    from swingnn_amd import engine as E
    c = MERGE_FUSED_CASES[name]
    sd = H.merge_state(c["C"], c["seed"])
    m = load(S.PatchMerging((c["res"],) * 2, c["C"]), strip(sd, "m."))
    x = filler.uniform("x", (c["B"], c["res"] ** 2, c["C"]), c["seed"])
    assert c["B"] * (c["res"] // 2) ** 2 >= 16384 and c["C"] in E.MERGE_FUSED_NC
    out = m(x.to(DEV)).cpu()
    ref = O.patch_merging(x, sd, "m", c["res"])
    rel_close(out, ref)
    saved = E.USE_FUSED_MLP
    E.USE_FUSED_MLP = False          # no weight image in the descriptor: LayerNorm pass + GEMM
    try:
        out2 = m(x.to(DEV)).cpu()
    finally:
        E.USE_FUSED_MLP = saved
    rel_close(out2, ref)
    # both round the same LayerNorm output to bf16 and accumulate in fp32: they differ by summation order only
    d = (out - out2).abs()
    scale = float(ref.abs().max())
    assert float(d.max()) <= 2e-3 * scale and float(d.mean()) <= 5e-5 * scale, (float(d.max()), float(d.mean()), scale)


# PatchBreakup (with the concat of :743) at token counts where the one-pass kernel (csrc/expand_fused.hip) runs:
# Cin = 384 / 768, >= 16384 input tokens.  Ragged: token count not a multiple of the 128-token workgroup, odd side.
EXPAND_FUSED_CASES = {
    "c384_r48_b8": dict(dim=384, res=48, B=8, seed=75),
    "c384_r45_b9": dict(dim=384, res=45, B=9, seed=76),
    "c768_r24_b32": dict(dim=768, res=24, B=32, seed=77),
    "c768_r25_b27": dict(dim=768, res=25, B=27, seed=78),
}


@pytest.mark.parametrize("name", list(EXPAND_FUSED_CASES))
def test_patch_breakup_one_pass_kernel_matches_oracle_and_gemm_path(name):
    import swingnn_amd as S
    from swingnn_amd import engine as E
    c = EXPAND_FUSED_CASES[name]
    dim = c["dim"]
    sd = H.breakup_state(dim, c["seed"])
    u = load(S.PatchBreakup((c["res"],) * 2, dim), strip(sd, "u."))
    assert c["B"] * c["res"] ** 2 >= 16384 and dim in E.EXPAND_FUSED_DIMS
    xb = filler.uniform("xb", (c["B"], c["res"] ** 2, dim), c["seed"])
    out = u(xb.to(DEV)).cpu()
    ref = O.patch_breakup(xb, sd, "u", c["res"])
    rel_close(out, ref)
    saved = E.USE_FUSED_MLP
    E.USE_FUSED_MLP = False          # no weight images in the descriptor: concat pass, GEMM, breakup tail kernel
    try:
        out2 = u(xb.to(DEV)).cpu()
    finally:
        E.USE_FUSED_MLP = saved
    rel_close(out2, ref)
    # the one-pass kernel normalises the fp32 accumulators, the other path the fp32 GEMM output: same roundings up to
    # summation order, amplified by the two LayerNorms
    d = (out - out2).abs()
    scale = float(ref.abs().max())
    assert float(d.max()) <= 1e-2 * scale and float(d.mean()) <= 2e-4 * scale, (float(d.max()), float(d.mean()), scale)


def test_patch_breakup_in_place_call_takes_the_safe_path():
    """swg_patch_expand_fwd with x_out starting at x (the way a caller with one token tensor would use it): the
    one-pass kernel cannot run in place (a workgroup's output rows are other workgroups' input rows), so the library
    must fall back to the concat + GEMM + tail route, which reads x completely before anything is written."""
    import ctypes as C
    import swingnn_amd as S
    from swingnn_amd import _lib as L, engine as E
    dim, res, B, seed = 384, 48, 8, 79
    sd = H.breakup_state(dim, seed)
    u = load(S.PatchBreakup((res,) * 2, dim), strip(sd, "u."))
    xb = filler.uniform("xb", (B, res ** 2, dim), seed)
    ref = O.patch_breakup(xb, sd, "u", res)
    T = B * res * res
    x, skip = [t.contiguous().to(DEV) for t in xb.chunk(2, dim=-1)]
    buf = torch.empty(T * dim, dtype=torch.float32, device=DEV)     # out [4T, dim/4]; x occupies its first half
    buf[:T * dim // 2].copy_(x.reshape(-1))
    keep = E.Keep()
    desc = E.breakup_desc(keep, u)
    assert desc.img_pre and desc.img_post
    ws = torch.empty(64 * x.numel() * 4 + (1 << 20), dtype=torch.uint8, device=DEV)
    L.check(L.lib.swg_patch_expand_fwd(C.byref(desc), B, buf.data_ptr(), skip.data_ptr(), buf.data_ptr(), ws.data_ptr(),
                                       ws.numel(), E.stream_ptr()), "swg_patch_expand_fwd")
    rel_close(buf.reshape(B, 4 * res * res, dim // 4), ref)


# C=768 with UN-clamped windows (the last stage of the full-size nets: res 12 / w 6 and res 16 / w 8, 4 windows per
# graph), which no reduced golden geometry reaches; shifted variants exercise the mask path of that kernel route.
BIG_BLOCK_CASES = {
    "r12_w6_c768": dict(C=768, res=12, heads=24, window=6, shift=0, B=3, seed=81),
    "r16_w8_c768": dict(C=768, res=16, heads=24, window=8, shift=0, B=2, seed=82),
    "r12_w6_s3_c768": dict(C=768, res=12, heads=24, window=6, shift=3, B=2, seed=83),
    "r24_w6_s3_c384": dict(C=384, res=24, heads=12, window=6, shift=3, B=2, seed=84),   # 16 shifted windows (grid s2)
    "r32_w8_s4_c384": dict(C=384, res=32, heads=12, window=8, shift=4, B=1, seed=85),   # DD s2
    "r48_w6_c192": dict(C=192, res=48, heads=6, window=6, shift=0, B=1, seed=86),       # grid s1: 64 windows
}


@pytest.mark.parametrize("name", list(BIG_BLOCK_CASES))
def test_block_full_size(name):
    import swingnn_amd as S
    c = BIG_BLOCK_CASES[name]
    w, s = O.eff_window(c["res"], c["window"], c["shift"])
    assert (w, s) == (c["window"], c["shift"])
    for regime in ("stress", "init"):
        full = H.block_state(c["C"], c["heads"], w, c["seed"], regime=regime)
        blk = load(S.SwinTransformerBlock(c["C"], 512, (c["res"],) * 2, c["heads"], c["window"], c["shift"]),
                   strip(full, "blk."))
        x = filler.uniform("x", (c["B"], c["res"] ** 2, c["C"]), c["seed"])
        emb = filler.uniform("emb", (c["B"], 512), c["seed"])
        out = blk(x.to(DEV), emb.to(DEV))
        ref = O.swin_block(x, emb, full, "blk", c["res"], c["heads"], c["window"], c["shift"])
        if regime == "stress":
            rel_close(out, ref)
        else:
            rel_close(out, ref, 5e-3, 5e-4)


@pytest.mark.parametrize("shift,in_place", [(0, False), (3, True)])
def test_block_fused_tail_matches_two_kernel_path(shift, in_place):
    """C = 384 at the bench's token count per launch (56 graphs x 24 x 24 tokens): swg_block_fwd runs the projection,
    the shortcut and the MLP half as ONE kernel behind the attention core (mlp_stream.hip, PROJ form).  It must agree
    with swg_wattn_fwd + swg_ln_mlp_fwd (projection GEMM, separate MLP kernel: same bf16 operands, fp32 accumulation in
    another order), and -- on a few graphs -- with the oracle to the block bar."""
    import ctypes as C
    import swingnn_amd as S
    from swingnn_amd import _lib as L, engine as E
    Cd, res, heads, win, B = 384, 24, 12, 6, 56
    full = H.block_state(Cd, heads, win, 91, regime="stress")
    blk = load(S.SwinTransformerBlock(Cd, 512, (res, res), heads, win, shift), strip(full, "blk."))
    x = filler.uniform("x", (B, res * res, Cd), 91)
    emb = filler.uniform("emb", (B, 512), 91)
    xd = x.to(DEV)
    cond = torch.addmm(blk.affine.bias, emb.to(DEV), blk.affine.weight.t()).contiguous()
    keep = E.Keep()
    d = E.block_desc(keep, blk, 0)
    assert d.proj_mlp_img, "the block must carry the combined projection + MLP image"
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    two = torch.empty_like(xd)
    L.check(L.lib.swg_wattn_fwd(C.byref(d), B, xd.data_ptr(), cond.data_ptr(), 2 * Cd, two.data_ptr(), ws.data_ptr(),
                                ws.numel(), E.stream_ptr()), "wattn")
    L.check(L.lib.swg_ln_mlp_fwd(C.byref(d), B, two.data_ptr(), two.data_ptr(), ws.data_ptr(), ws.numel(),
                                 E.stream_ptr()), "mlp")
    buf = xd.clone() if in_place else torch.empty_like(xd)
    src = buf if in_place else xd
    L.check(L.lib.swg_block_fwd(C.byref(d), B, src.data_ptr(), cond.data_ptr(), 2 * Cd, buf.data_ptr(), ws.data_ptr(),
                                ws.numel(), E.stream_ptr()), "block")
    torch.cuda.synchronize()
    # the fp32 attention-half result differs in its last bits (summation order), which flips some bf16 roundings of the
    # LN2 output: isolated entries move by a bf16 ulp of an operand, the bulk agrees to fp32 rounding
    scale = float(two.abs().max())
    diff = (buf - two).abs()
    assert float(diff.max()) <= 3e-3 * scale, float(diff.max()) / scale
    assert float(diff.mean()) <= 5e-5 * scale, float(diff.mean()) / scale
    ref = O.swin_block(x[:2], emb[:2], full, "blk", res, heads, win, shift)
    rel_close(buf[:2].cpu(), ref)


@pytest.fixture(scope="module")
def golden_long():
    path = os.path.join(os.path.dirname(__file__), "golden", "golden_sampler_long.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", list(SAMPLER_LONG_CASES))
def test_sampler_256_steps(golden_long, golden_meta, name):
    """bf16 drift over the whole trajectory (766 network passes) against the reference's own sample."""
    import swingnn_amd as S
    from tests.test_gpu_parity import build_precond
    c = SAMPLER_LONG_CASES[name]
    cfg = NET_CASES[c["net"]]["cfg"]
    pre = build_precond(S, c["net"], golden_meta, c["regime"], seed=c["seed"])
    smp = S.EDMSampler(num_steps=c["steps"], clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                       clip_samples_scope="x_0", self_condition=cfg["self_condition"], dev=DEV)
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    smp._randn_like = lambda x: torch.randn(x.shape).to(x.device)   # the reference drew from the CPU generator
    final, interim = smp.sample(pre, flags.to(DEV), flag_interim_adjs=True, max_num_interim_adjs=5)
    ref = torch.from_numpy(golden_long[f"sampler/{name}/final"])
    assert interim.shape == golden_long[f"sampler/{name}/interim"].shape
    err = float((final - ref).abs().max())
    thr = O.threshold_adj(final, flags)
    thr_ref = torch.from_numpy(golden_long[f"sampler/{name}/thresholded"])
    valid = (flags[:, :, None] * flags[:, None, :]) > 0
    raw = float((thr == thr_ref)[valid].float().mean())
    sure = valid & (ref.abs() > 1e-2)
    filt = float((thr == thr_ref)[sure].float().mean())
    print(f"{name}: final max err {err:.3g}, thresholded agreement raw {raw:.5f} filtered {filt:.5f}")
    assert err <= 1e-2, err
    assert filt >= 0.999 and raw >= 0.999, (raw, filt)


@pytest.mark.parametrize("name", list(SAMPLER_FULL_CASES))
def test_sampler_full_geometry(name):
    """EDMSampler.sample at the EXACT bench geometries (grid N=384 and DD N=512: 8 Heun steps, B = 1) against the
    reference's own trajectory (golden_sampler_full.npz, made by tests/golden/make_golden_sampler_long.py --only full
    from runner/mcmc_sampler/edm.py:125-232 with the same seeded CPU draws)."""
    import swingnn_amd as S
    c = SAMPLER_FULL_CASES[name]
    cfg = c["cfg"]
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_sampler_full.npz"), allow_pickle=False) as z:
        ref, thr_ref = torch.from_numpy(z[f"sampler/{name}/final"]), torch.from_numpy(z[f"sampler/{name}/thresholded"])
    net = S.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=1, embed_dim=cfg["embed_dim"],
                    depths=cfg["depths"], num_heads=cfg["num_heads"], window_size=cfg["window_size"], mlp_ratio=4.,
                    drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0, self_condition=True)
    pre = S.Precond("edm", net, True)
    pre.load_state_dict(filler.fill_state_dict(pre.state_dict(), c["seed"], c["regime"]))
    pre = pre.to(DEV).eval()
    smp = S.EDMSampler(num_steps=c["steps"], clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                       clip_samples_scope="x_0", self_condition=True, dev=DEV)
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    smp._randn_like = lambda x: torch.randn(x.shape).to(x.device)   # the reference drew from the CPU generator
    final = smp.sample(pre, flags.to(DEV))
    err = float((final - ref).abs().max())
    thr = O.threshold_adj(final, flags)
    valid = (flags[:, :, None] * flags[:, None, :]) > 0
    raw = float((thr == thr_ref.float())[valid].float().mean())
    sure = valid & (ref.abs() > 1e-2)
    filt = float((thr == thr_ref.float())[sure].float().mean())
    print(f"{name}: final max err {err:.3g}, thresholded agreement raw {raw:.5f} filtered {filt:.5f}")
    assert err <= 1e-2, err
    assert filt >= 0.999 and raw >= 0.999, (raw, filt)
