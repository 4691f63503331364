"""Host-side data formats either side of the hot path (swingnn_amd/io.py) against the reference's own helpers
(goldens: tests/golden/make_golden_data.py; the thresholded samples of tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden.cases import NET_CASES, SAMPLER_CASES
from oracle import filler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dataset_tensorisation_matches_reference():
    from swingnn_amd import io
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_data.npz"), allow_pickle=False) as z:
        ref_adjs, ref_flags = z["adjs"], z["flags"]
    adjs, feat, flags = io.adjs_to_dataset_tensors(H.synthetic_graph_list(), 20, ddpm_scale=True)
    assert np.array_equal(adjs.numpy(), ref_adjs) and np.array_equal(flags.numpy(), ref_flags)
    assert feat.shape == (5, 20, 1) and float(feat.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        io.pad_adjs(np.zeros((21, 21)), 20)


@pytest.mark.parametrize("name", [k for k, c in SAMPLER_CASES.items() if not c["sanity"]])
def test_quantisation_matches_reference(golden, name):
    from swingnn_amd import io
    c = SAMPLER_CASES[name]
    cfg = NET_CASES[c["net"]]["cfg"]
    flags = filler.node_flags("flags", c["B"], cfg["img_size"], c["n_lo"], c["n_hi"], c["seed"])
    q = io.quantise_samples(torch.from_numpy(golden[f"sampler/{name}/final"]), flags)
    assert np.array_equal(q.numpy(), golden[f"sampler/{name}/thresholded"])


def test_load_model_reconciles_module_prefix(golden_meta):
    import swingnn_amd as S
    from swingnn_amd import io
    cfg = NET_CASES["ego60"]["cfg"]

    def make():
        net = S.SwinGNN(img_size=cfg["img_size"], patch_size=cfg["patch_size"], in_chans=1, embed_dim=cfg["embed_dim"],
                        depths=cfg["depths"], num_heads=cfg["num_heads"], window_size=cfg["window_size"], mlp_ratio=4.,
                        drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0, self_condition=True)
        return S.Precond("edm", net, True)

    src, dst = make(), make()
    sd = H.net_state(golden_meta["keys/ego60"], 9, "init")
    src.load_state_dict({k: v for k, v in sd.items() if k in src.state_dict()}, strict=False)
    ddp_ckpt = {"model": {"module." + k: v.clone() for k, v in src.state_dict().items()}}    # saved under DDP
    io.load_model(ddp_ckpt, dst, "model")
    for (k0, v0), (k1, v1) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert k0 == k1 and torch.equal(v0, v1)
    plain = {"model_ema_beta_0.9999": {k: v.clone() for k, v in src.state_dict().items()}}
    io.load_model(plain, make(), "model_ema_beta_0.9999")
    with pytest.raises(AssertionError):
        io.load_model(plain, dst, "model")


def test_tile_batch_matches_trainer_rule():
    """runner/trainer/trainer.py:29-47: short batches are stacked to batch_size and cached on the dataloader."""
    from swingnn_amd import io

    class DL:
        batch_size = 12

    dl = DL()
    adjs = torch.arange(4 * 5 * 5, dtype=torch.float32).reshape(4, 5, 5)
    adjs[1] = 0                                                   # an empty graph: all flags 0
    x = torch.zeros(4, 5, 1)
    a, xx, f = io.tile_batch((adjs, x), dl)
    assert a.shape == (12, 5, 5) and xx.shape == (12, 5, 1) and f.shape == (12, 5)
    assert torch.equal(a, adjs.repeat(3, 1, 1)) and torch.equal(f[:4], adjs.abs().sum(-1).gt(1e-5).float())
    assert f[1].sum() == 0 and dl.repeated_data[0] is a
    # the cache wins for every later short batch (the reference's behaviour, even for different data)
    other = torch.ones(4, 5, 5)
    a2, _, f2 = io.tile_batch((other, x, torch.ones(4, 5)), dl)
    assert a2 is a and f2 is f
    # full batches and non-dividing sizes pass through untouched
    dl2 = DL()
    full = torch.ones(12, 5, 5)
    assert io.tile_batch((full, torch.zeros(12, 5, 1), torch.ones(12, 5)), dl2)[0] is full
    odd = torch.ones(5, 5, 5)
    assert io.tile_batch((odd, torch.zeros(5, 5, 1), torch.ones(5, 5)), dl2)[0] is odd and not hasattr(dl2, "repeated_data")
