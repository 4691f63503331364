"""GPU tests of the host-side launch planning around the hot path (swingnn_amd/engine.py): HIP-graph scratch lifetime,
the part-batch split against the one-pass kernels' token gates, and the float4 forms of the public elementwise entry
points on misaligned views.  Everything is compared with the eager / aligned call of the same entry point."""
import numpy as np
import pytest
import torch

from oracle import filler
from tests.golden.cases import HEADS, NET_CASES
from tests.test_gpu_parity import DEV, build_precond

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_recorded_graph_survives_workspace_growth(golden_meta, monkeypatch):
    """A recorded pass bakes the address of its scratch into the graph nodes.  The eager pool replaces a buffer when a
    later, larger pass outgrows it; the graph must keep replaying correctly afterwards (it owns its scratch), both for
    the one-stream recording and for the recording with two part-batch branches."""
    import swingnn_amd as S
    from swingnn_amd import engine as E
    for net_name, B1, B2, split in (("ego60", 3, 64, False), ("grid192", 2, 8, True)):
        monkeypatch.setattr(E, "SPLIT_MIN_WORK", 0 if split else 1 << 60)
        nc = NET_CASES[net_name]
        cfg = nc["cfg"]
        N = cfg["img_size"]
        pre = build_precond(S, net_name, golden_meta, "init", seed=7)
        mk = lambda B, seed: (filler.node_flags("flags", B, N, nc["n_lo"], nc["n_hi"], seed).to(DEV),)
        fl1, = mk(B1, 5)
        x1 = filler.sym_adj("x", B1, N, fl1.cpu(), 5, 2.0).to(DEV)
        fl2, = mk(B2, 6)
        x2 = filler.sym_adj("x", B2, N, fl2.cpu(), 6, 2.0).to(DEV)
        sig = torch.full((1,), 0.9, device=DEV)
        monkeypatch.setattr(np.random, "rand", lambda: 0.75)      # one network pass per call
        eager1 = pre(x1, fl1, sig.expand(B1)).clone()
        E._WS.clear()                                             # the pool starts small: B1 sizes it
        pre.use_graphs = True
        rec = pre(x1, fl1, sig.expand(B1)).clone()                # records (and replays once)
        assert torch.equal(rec, eager1)
        pre.use_graphs = False
        big = pre(x2, fl2, sig.expand(B2)).clone()                # eager pass that outgrows every pooled buffer
        junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(8)]   # recycle what the pool released
        torch.cuda.synchronize()
        pre.use_graphs = True
        r0 = E.GRAPH_REPLAYS
        again = pre(x1, fl1, sig.expand(B1)).clone()              # replay of the B1 graph
        assert E.GRAPH_REPLAYS == r0 + 1
        assert torch.equal(again, eager1), float((again - eager1).abs().max())
        pre.use_graphs = False
        assert torch.equal(pre(x2, fl2, sig.expand(B2)), big)
        del junk
        ent = next(iter(E.get_cache(pre.model)._graphs.values()))
        assert len(ent) == 4 and all(t.numel() >= E.get_cache(pre.model).ws_bytes(B1 // (2 if split else 1)) for t in ent[3])


def test_split_never_drops_a_stage_below_a_fused_gate(monkeypatch):
    """grid N=384 dim-96: a pass is issued as two part-batches only when the part-batch takes the same one-pass kernels
    as the whole batch (token gate 12288: 22 graphs at C = 384).  Round 2 split B = 56 into 2 x 28 graphs, under its
    gate of 16384 tokens, and the C = 384 stage silently fell back to the GEMM route."""
    import swingnn_amd as S
    from swingnn_amd import engine as E
    net = S.SwinGNN(img_size=384, patch_size=4, in_chans=1, embed_dim=96, depths=[1, 1, 3, 1], num_heads=HEADS,
                    window_size=6, mlp_ratio=4., drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0,
                    self_condition=True).to(DEV).eval()
    cache = E.get_cache(net)
    full = cache.fused_ops(112)
    assert full > 0 and cache.fused_ops(56) == full and cache.fused_ops(28) == full and cache.fused_ops(20) < full
    assert E.launch_batch(112, 384, cache) == 56
    assert E.launch_batch(56, 384, cache) == 28      # 28 x 576 = 16128 tokens: above the gate, same kernels
    assert E.launch_batch(57, 384, cache) == 57      # odd batches never split
    assert E.launch_batch(8, 384, cache) == 8        # small work stays on one stream
    monkeypatch.setattr(E, "SPLIT_MIN_WORK", 0)
    assert E.launch_batch(40, 384, cache) == 40      # 2 x 20 graphs = 11520 tokens would drop under the gate: refused
    assert E.launch_batch(44, 384, cache) == 22      # 22 x 576 = 12672: allowed
    monkeypatch.undo()
    # the pass itself: bit-identical whichever way it is launched
    B = 56
    flags = (torch.arange(384)[None, :] < 361).float().expand(B, -1).contiguous().to(DEV)
    x = torch.randn(B, 384, 384, device=DEV)
    x = (x + x.transpose(1, 2)) * flags[:, :, None] * flags[:, None, :]
    lab = torch.full((1,), 0.3, device=DEV).expand(B)
    a = net(x, flags, lab)
    E_SPLIT = E.SPLIT_STREAMS
    try:
        E.SPLIT_STREAMS = 0
        b = net(x, flags, lab)
    finally:
        E.SPLIT_STREAMS = E_SPLIT
    assert torch.equal(a, b)


def test_float4_forms_fall_back_on_misaligned_views():
    """swg_heun_euler / swg_heun_correct / the symmetrise epilogue take arbitrary float*: a view that starts 4 bytes into
    an allocation (N % 4 == 0 still holds) takes the scalar form and must agree with the aligned (float4) call to fp32
    rounding (the two forms contract their multiply-adds differently)."""
    import swingnn_amd._lib as L
    from swingnn_amd import engine as E
    B, N = 3, 24
    flags = filler.node_flags("f", B, N, 4, 20, 1).to(DEV)
    mk = lambda name, seed: filler.sym_adj(name, B, N, flags.cpu(), seed, 2.0).to(DEV)
    xh, den, den2 = mk("xh", 1), mk("den", 2), mk("den2", 3)
    sp = E.stream_ptr()

    def off(t):            # same values, data pointer 4 bytes past a 16-byte boundary
        buf = torch.empty(t.numel() + 1, device=DEV)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v

    def run(conv):
        a, b, c, f = conv(xh), conv(den), conv(den2), conv(flags)
        dc, xp, xn = conv(torch.zeros_like(xh)), conv(torch.zeros_like(xh)), conv(torch.zeros_like(xh))
        L.check(L.lib.swg_heun_euler(B, N, a.data_ptr(), b.data_ptr(), f.data_ptr(), 2.5, -0.4, 0, dc.data_ptr(),
                                     xp.data_ptr(), sp), "euler")
        L.check(L.lib.swg_heun_correct(B, N, a.data_ptr(), xp.data_ptr(), dc.data_ptr(), c.data_ptr(), f.data_ptr(),
                                       2.1, -0.4, xn.data_ptr(), sp), "correct")
        return dc.clone(), xp.clone(), xn.clone()

    for got, want in zip(run(off), run(lambda t: t.contiguous())):
        assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
