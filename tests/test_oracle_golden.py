"""Pin the CPU oracle against golden vectors produced by the reference's own modules.

fp32 vs fp32 on the same CPU: tolerance 2e-5 relative to the output scale (different op
fusion / reduction order only)."""
import numpy as np
import pytest
import torch

from oracle import filler
from oracle import swingnn_ref as O
from tests import helpers as H
from tests.golden.cases import (ATTN_CASES, BLOCK_CASES, EMBED_CASES, MERGE_CASES, NET_CASES, SAMPLER_CASES)

torch.set_grad_enabled(False)


def close(a, ref, rtol=2e-5):
    ref = torch.as_tensor(ref)
    scale = max(1.0, float(ref.abs().max()))
    err = H.max_abs(a, ref)
    assert err <= rtol * scale, f"max abs err {err:.3e} (scale {scale:.3g})"


@pytest.mark.parametrize("name", list(ATTN_CASES))
def test_window_attention(golden, name):
    c = ATTN_CASES[name]
    sd = H.attn_state(c["C"], c["heads"], c["w"], c["seed"])
    x = filler.uniform("x", (c["B_"], c["w"] ** 2, c["C"]), c["seed"])
    mask = O.shift_mask(c["mask_res"], c["w"], c["w"] // 2) if c["mask_res"] else None
    close(O.window_attention(x, sd, "attn", c["heads"], c["w"], mask), golden[f"attn/{name}"])


@pytest.mark.parametrize("name", list(BLOCK_CASES))
def test_swin_block(golden, name):
    c = BLOCK_CASES[name]
    w, s = O.eff_window(c["res"], c["window"], c["shift"])
    # buffers recomputed by the oracle must equal the reference's registered buffers
    assert np.array_equal(O.relative_position_index(w).numpy(), golden[f"block/{name}/rel_index"])
    if s > 0:
        assert np.array_equal(O.shift_mask(c["res"], w, s).numpy(), golden[f"block/{name}/attn_mask"])
    else:
        assert f"block/{name}/attn_mask" not in golden
    sd = H.block_state(c["C"], c["heads"], w, c["seed"])
    x = filler.uniform("x", (c["B"], c["res"] ** 2, c["C"]), c["seed"])
    emb = filler.uniform("emb", (c["B"], 512), c["seed"])
    close(O.swin_block(x, emb, sd, "blk", c["res"], c["heads"], c["window"], c["shift"]), golden[f"block/{name}/out"])


@pytest.mark.parametrize("name", list(MERGE_CASES))
def test_merge_breakup(golden, name):
    c = MERGE_CASES[name]
    x = filler.uniform("x", (c["B"], c["res"] ** 2, c["C"]), c["seed"])
    close(O.patch_merging(x, H.merge_state(c["C"], c["seed"]), "m", c["res"]), golden[f"merge/{name}"])
    xb = filler.uniform("xb", (c["B"], (c["res"] // 2) ** 2, 4 * c["C"]), c["seed"])
    close(O.patch_breakup(xb, H.breakup_state(4 * c["C"], c["seed"]), "u", c["res"] // 2), golden[f"breakup/{name}"])


@pytest.mark.parametrize("name", list(EMBED_CASES))
def test_patch_embed(golden, name):
    c = EMBED_CASES[name]
    x = filler.uniform("x", (c["B"], c["in"], c["N"], c["N"]), c["seed"])
    emb = filler.uniform("emb", (c["B"], 512), c["seed"])
    close(O.patch_embed(x, emb, H.embed_state(c["C"], c["in"], c["p"], c["seed"]), "pe", c["p"]), golden[f"embed/{name}"])


@pytest.mark.parametrize("name", list(NET_CASES))
def test_network_and_precond(golden, golden_meta, name):
    c = NET_CASES[name]
    cfg = c["cfg"]
    for regime in c["regimes"]:
        sd = H.net_state(golden_meta[f"keys/{name}"], c["seed"], regime)
        net = O._strip(sd)
        for si in range(len(c["sigmas"])):
            flags, x, sig, sc = H.net_inputs(c, si)
            c_noise = sig.log() / 4
            close(O.swingnn_forward(net, cfg, x, flags, c_noise, None), golden[f"net/{name}/{regime}/s{si}/F_nosc"], 5e-5)
            close(O.swingnn_forward(net, cfg, x, flags, c_noise, sc), golden[f"net/{name}/{regime}/s{si}/F_sc"], 5e-5)
            close(O.precond_forward(sd, cfg, x, flags, sig, True), golden[f"net/{name}/{regime}/s{si}/D_coin1"], 5e-5)
            close(O.precond_forward(sd, cfg, x, flags, sig, False), golden[f"net/{name}/{regime}/s{si}/D_coin0"], 5e-5)


def test_collapsed_read_out_matches_three_convs(golden_meta):
    c = NET_CASES["grid192"]
    sd = O._strip(H.net_state(golden_meta["keys/grid192"], 7, "stress"))
    p, C = 4, 96
    tok = filler.uniform("tok", (2, 36, C), 7)
    full = O.read_out(tok, sd, 6, p)
    Weff, beff = O.collapsed_read_out(sd, p)
    y = (tok.double() @ Weff + beff).reshape(2, 6, 6, p, p).permute(0, 1, 3, 2, 4).reshape(2, 24, 24)
    close(y.float(), full, 1e-5)


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
def test_edm_sampler(golden, golden_meta, name):
    c = SAMPLER_CASES[name]
    nc = NET_CASES[c["net"]]
    cfg = nc["cfg"]
    assert np.array_equal(O.edm_sigma_steps(c["steps"])[:-1].numpy(),
                          golden[f"sampler/{name}/sigma_steps"].astype(np.float32))
    sd = H.net_state(golden_meta[f"keys/{c['net']}"], c["seed"], c["regime"])
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    gt = filler.sym_adj("gt", c["B"], cfg["img_size"], flags, c["seed"]).sign() if c["sanity"] else None
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])

    def denoiser(x, fl, sig, sc):
        return O.precond_forward(sd, cfg, x, fl, sig, np.random.rand() < 0.5)

    out = O.edm_sample(denoiser, flags, num_steps=c["steps"], self_condition=cfg["self_condition"],
                       sanity_check_gt_adjs=gt)
    ref = torch.from_numpy(golden[f"sampler/{name}/final"])
    close(out, ref, 2e-4 if c["steps"] > 64 else 5e-5)
    thr = O.threshold_adj(out, flags)
    agree = float((thr == torch.from_numpy(golden[f"sampler/{name}/thresholded"])).float().mean())
    assert agree >= 0.999, agree
    if c["sanity"]:
        # known-answer mode (trainer.py:222-224): a perfect denoiser reproduces the ground truth
        assert H.max_abs(out, gt) < 1e-5


def test_edm_sampler_long_com96(golden_meta):
    """The oracle replays the reference's full 256-step community-small dim-96 sample (golden_sampler_long.npz; the
    grid192 trajectory of that file costs a minute of CPU and is replayed by the GPU test only)."""
    import os
    from tests.golden.cases import SAMPLER_LONG_CASES
    name = "com96_256"
    c = SAMPLER_LONG_CASES[name]
    cfg = NET_CASES[c["net"]]["cfg"]
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_sampler_long.npz"), allow_pickle=False) as z:
        ref, thr_ref = torch.from_numpy(z[f"sampler/{name}/final"]), torch.from_numpy(z[f"sampler/{name}/thresholded"])
    sd = H.net_state(golden_meta[f"keys/{c['net']}"], c["seed"], c["regime"])
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    out = O.edm_sample(lambda x, fl, sig, sc: O.precond_forward(sd, cfg, x, fl, sig, np.random.rand() < 0.5), flags,
                       num_steps=c["steps"], self_condition=True)
    close(out, ref, 2e-4)
    assert float((O.threshold_adj(out, flags) == thr_ref).float().mean()) >= 0.999


def test_edm_sampler_full_geometry_grid384():
    """The oracle replays the reference's 8-step EDMSampler.sample at the FULL grid geometry (N = 384, dim-96,
    golden_sampler_full.npz from make_golden_sampler_long.py --only full): pins the oracle where bench.py runs."""
    import os
    from tests.golden.cases import SAMPLER_FULL_CASES
    name = "grid384_8"
    c = SAMPLER_FULL_CASES[name]
    cfg = c["cfg"]
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_sampler_full.npz"), allow_pickle=False) as z:
        ref, thr_ref = torch.from_numpy(z[f"sampler/{name}/final"]), torch.from_numpy(z[f"sampler/{name}/thresholded"])
    import swingnn_amd as S      # parameter containers only (same state_dict keys / shapes as the reference's modules)
    net = S.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=1, embed_dim=cfg["embed_dim"],
                    depths=cfg["depths"], num_heads=cfg["num_heads"], window_size=cfg["window_size"], mlp_ratio=4.,
                    drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0, self_condition=True)
    sd = filler.fill_state_dict(S.Precond("edm", net, True).state_dict(), c["seed"], c["regime"])
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    torch.manual_seed(c["seed"])
    np.random.seed(c["seed"])
    with torch.no_grad():
        out = O.edm_sample(lambda x, fl, sig, sc: O.precond_forward(sd, cfg, x, fl, sig, np.random.rand() < 0.5), flags,
                           num_steps=c["steps"], self_condition=True)
    close(out, ref, 5e-5)
    assert float((O.threshold_adj(out, flags) == thr_ref.float()).float().mean()) >= 0.9999


def test_objective_and_loss(golden):
    B, N = 6, 24
    flags = filler.node_flags("flags", B, N, 5, 20, 3)
    clean = filler.sym_adj("clean", B, N, flags, 3).sign()
    torch.manual_seed(5)
    sig, w = O.edm_training_sigmas_weights(torch.randn(B))
    noisy = O.edm_noisy_input(clean, flags, sig, torch.randn_like(clean))
    close(sig, golden["objective/sigmas"])
    close(w, golden["objective/weights"])
    close(noisy, golden["objective/noisy"])
    pred = filler.sym_adj("pred", B, N, flags, 4)
    close(O.edm_loss(pred, clean, flags, w), golden["objective/loss"])


# ---- G9: train step (row a20): autograd through the oracle against the reference's loss.backward() ----
from tests.golden.cases import TRAIN_CASES  # noqa: E402


@pytest.mark.parametrize("name,regime,si,coin", TRAIN_CASES)
def test_oracle_train_step_gradients(golden_meta, golden_train, name, regime, si, coin):
    arrays, meta = golden_train
    c = NET_CASES[name]
    sd = H.net_state(golden_meta[f"keys/{name}"], c["seed"], regime)
    flags, x, sig, w, clean = H.train_case_inputs(c, si)
    loss, grads = H.oracle_grads(sd, c["cfg"], x, flags, sig, w, clean, coin)
    key = f"train/{name}/{regime}/s{si}/coin{int(coin)}"
    np.testing.assert_allclose(loss.numpy(), arrays[key + "/loss"], rtol=2e-5, atol=1e-7)
    names = meta["param_names/" + name]
    g = [grads[k].double() for k in names]
    total = float(torch.sqrt(sum((t ** 2).sum() for t in g)))
    assert abs(total - float(arrays[key + "/grad_norm"])) <= 1e-4 * total
    abs_sums = np.array([float(t.abs().sum()) for t in g])
    sums = np.array([float(t.sum()) for t in g])
    ref_abs = arrays[key + "/grad_abs_sums"]
    np.testing.assert_allclose(abs_sums, ref_abs, rtol=2e-3, atol=1e-6 * float(ref_abs.max()))
    np.testing.assert_allclose(sums, arrays[key + "/grad_sums"], rtol=0, atol=2e-3 * float(ref_abs.max()))
    for k in names:
        gk = f"{key}/grad/{k}"
        if gk in arrays:
            ref = arrays[gk]
            err = np.abs(grads[k].numpy() - ref).max()
            assert err <= 1e-3 * max(np.abs(ref).max(), 1e-6), (k, err)
