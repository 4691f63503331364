"""Case tables shared by the golden generator and the parity tests (data only)."""

HEADS = [3, 6, 12, 24]


def _cfg(img, p, dim, depths, window, self_cond=True):
    return dict(img_size=img, patch_size=p, in_chans=1, embed_dim=dim, depths=depths, num_heads=HEADS,
                window_size=window, self_condition=self_cond)


# G1: WindowAttention.  mask_res != 0 -> use the shift mask of a res x res block (B_ must be k * nW)
ATTN_CASES = {
    "w6_c96": dict(C=96, w=6, heads=3, B_=8, mask_res=0, seed=11),
    "w6_c96_mask": dict(C=96, w=6, heads=3, B_=8, mask_res=12, seed=12),      # 4 windows x 2
    "w8_c192_mask": dict(C=192, w=8, heads=6, B_=4, mask_res=16, seed=13),
    "w4_c60": dict(C=60, w=4, heads=3, B_=3, mask_res=0, seed=14),            # head_dim 20
}

# G2/G3: SwinTransformerBlock
BLOCK_CASES = {
    "r12_w6_s0": dict(C=96, res=12, heads=3, window=6, shift=0, B=2, seed=21),
    "r12_w6_s3": dict(C=96, res=12, heads=3, window=6, shift=3, B=2, seed=22),
    "r18_w6_s3_c192": dict(C=192, res=18, heads=6, window=6, shift=3, B=1, seed=23),
    "r16_w8_s4": dict(C=96, res=16, heads=3, window=8, shift=4, B=2, seed=24),
    "r8_w24_clamp": dict(C=96, res=8, heads=3, window=24, shift=12, B=3, seed=25),   # clamp -> w=8, s=0
    "r2_w24_c240": dict(C=240, res=2, heads=12, window=24, shift=0, B=3, seed=26),   # 4-token window, hd 20
    "r6_w6_c768": dict(C=768, res=6, heads=24, window=6, shift=3, B=2, seed=27),     # res == w -> shift forced 0
}

# G4: PatchMerging (res, C) and PatchBreakup (res/2, dim=4C)
MERGE_CASES = {
    "r12_c96": dict(C=96, res=12, B=2, seed=31),
    "r8_c60": dict(C=60, res=8, B=2, seed=32),
    "r4_c384": dict(C=384, res=4, B=1, seed=33),
}

# G5: PatchEmbed
EMBED_CASES = {
    "p4_in2": dict(N=48, p=4, **{"in": 2}, C=96, B=2, seed=41),
    "p3_in2": dict(N=24, p=3, **{"in": 2}, C=60, B=2, seed=42),
    "p1_in24": dict(N=16, p=1, **{"in": 24}, C=96, B=2, seed=43),
}

# G6/G7: full SwinGNN / Precond.  regimes: see oracle/filler.py
NET_CASES = {
    # ego-small / community-small geometry (config/edm_swin_gnn/ego_small..., com_small...)
    "ego60": dict(cfg=_cfg(24, 3, 60, [4, 4, 6], 24), B=4, n_lo=4, n_hi=18, sigmas=[0.05, 1.0, 40.0],
                  regimes=["init", "stress"], seed=51),
    "com96": dict(cfg=_cfg(24, 3, 96, [4, 4, 6], 24), B=4, n_lo=12, n_hi=20, sigmas=[0.5, 10.0],
                  regimes=["init", "stress"], seed=52),
    # reduced grid geometry that still reaches the shifted path: token sides 48/24/12/6
    "grid192": dict(cfg=_cfg(192, 4, 96, [1, 1, 3, 1], 6), B=2, n_lo=100, n_hi=181, sigmas=[0.2, 5.0],
                    regimes=["init", "stress"], seed=53),
    # reduced DD geometry: token sides 64/32/16/8, w=8
    "dd256": dict(cfg=_cfg(256, 4, 96, [1, 1, 3, 1], 8), B=1, n_lo=130, n_hi=250, sigmas=[2.0],
                  regimes=["init"], seed=54),
    # no self-conditioning (in_chans stays 1)
    "ego60_nosc": dict(cfg=_cfg(24, 3, 60, [4, 4, 6], 24, self_cond=False), B=2, n_lo=4, n_hi=18, sigmas=[1.0],
                       regimes=["stress"], seed=55),
}

# G8: EDMSampler.sample (CPU torch generator + numpy coin, both seeded with `seed`)
SAMPLER_CASES = {
    "ego60_8": dict(net="ego60", regime="init", steps=8, B=3, n_lo=4, n_hi=18, sanity=False, seed=61),
    "ego60_256": dict(net="ego60", regime="init", steps=256, B=2, n_lo=4, n_hi=18, sanity=False, seed=62),
    "ego60_sanity": dict(net="ego60", regime="init", steps=16, B=3, n_lo=4, n_hi=18, sanity=True, seed=63),
    "grid192_4": dict(net="grid192", regime="init", steps=4, B=1, n_lo=100, n_hi=181, sanity=False, seed=64),
}

# Long trajectories (tests/golden/make_golden_sampler_long.py -> golden_sampler_long.npz): bf16 drift over all 256 Heun
# steps on the grid geometry (shifted 6x6 windows live) and on community-small dim-96.  GPU tests only read them; the
# CPU oracle test replays the cheap one.
SAMPLER_LONG_CASES = {
    "grid192_256": dict(net="grid192", regime="init", steps=256, B=1, n_lo=100, n_hi=181, sanity=False, seed=65),
    "com96_256": dict(net="com96", regime="init", steps=256, B=3, n_lo=12, n_hi=20, sanity=False, seed=66),
}

# Few-step trajectories at the FULL bench geometries (BASELINE.json configs[3] grid N=384 / configs[4] DD N=512, dim-96),
# B = 1: seconds of reference CPU each (a 4-step DD run was tried first: its three huge steps amplify bf16 rounding to 9e-3, at the edge of the 1e-2 bar) (runner/mcmc_sampler/edm.py:125-232).  Stored in golden_sampler_full.npz.
SAMPLER_FULL_CASES = {
    "grid384_8": dict(cfg=_cfg(384, 4, 96, [1, 1, 3, 1], 6), regime="init", steps=8, B=1, n_lo=361, n_hi=361, seed=67),
    "dd512_8": dict(cfg=_cfg(512, 4, 96, [1, 1, 3, 1], 8), regime="init", steps=8, B=1, n_lo=300, n_hi=500, seed=68),
}

# G9: train step (loss [B] + every parameter gradient) of RainbowLoss(Precond(SwinGNN)(x, flags, sigmas), sign(x)).mean()
# (network case, weight regime, sigma index, outcome of the self-conditioning coin)
TRAIN_CASES = [("com96", "stress", 0, True), ("com96", "init", 1, False), ("ego60", "stress", 1, True),
               ("grid192", "stress", 0, True), ("ego60_nosc", "stress", 0, False), ("grid192", "init", 1, True),
               ("ego60", "init", 0, True)]
# GPU-only extra (no stored fingerprint: gradients are checked against autograd through the oracle, whose forward for
# this geometry is pinned by the 'net/dd256/*' goldens): 8 x 8 windows WITH shift, the 64-token attention-backward path
TRAIN_CASES_GPU_EXTRA = [("dd256", "init", 0, True)]


# G10: NodeAdjSwinGNN / NodeAdjPrecond / NodeAdjEDMSampler (molecule path).  in_chans = 2*Cn + Ca before the
# self-conditioning doubling (utils/learning_utils.py:184-214); adjacency [B,N,N] when Ca == 1 else [B,Ca,N,N],
# node features [B,N] when Cn == 1 else [B,N,Cn].
def _na_cfg(img, p, dim, depths, window, Ca, Cn, self_cond=True):
    return dict(img_size=img, patch_size=p, in_chans=2 * Cn + Ca, embed_dim=dim, depths=depths, num_heads=HEADS,
                window_size=window, self_condition=self_cond, out_chans_adj=Ca, out_chans_node=Cn)


NODEADJ_CASES = {
    # config/edm_swin_gnn/qm9_edm_swin_gnn.yaml as shipped: ddpm encodings (1 + 1 channels), dim 60
    "qm9_ddpm60": dict(cfg=_na_cfg(16, 1, 60, [1, 1, 3, 1], 4, 1, 1), B=4, n_lo=3, n_hi=9, sigmas=[0.3, 8.0],
                       regimes=["init", "stress"], seed=71),
    # one-hot node / edge attributes (4 + 4 channels), dim 96 (BASELINE.json configs[2])
    "qm9_onehot96": dict(cfg=_na_cfg(16, 1, 96, [1, 1, 3, 1], 4, 4, 4), B=3, n_lo=3, n_hi=9, sigmas=[1.0],
                         regimes=["stress"], seed=72),
    # zinc250k geometry: N=48, patch 3, window 2 (token sides 16/8/4/2; the middle block of the depth-3 stage is shifted)
    "zinc_ddpm96": dict(cfg=_na_cfg(48, 3, 96, [1, 1, 3, 1], 2, 1, 1), B=2, n_lo=10, n_hi=38, sigmas=[2.0],
                        regimes=["stress"], seed=73),
    "qm9_ddpm60_nosc": dict(cfg=_na_cfg(16, 1, 60, [1, 1, 3, 1], 4, 1, 1, self_cond=False), B=2, n_lo=3, n_hi=9,
                            sigmas=[1.0], regimes=["stress"], seed=74),
}

NODEADJ_SAMPLER_CASES = {
    "qm9_ddpm60_6": dict(net="qm9_ddpm60", regime="init", steps=6, B=3, n_lo=3, n_hi=9, sanity=False, seed=81),
    "qm9_onehot96_4": dict(net="qm9_onehot96", regime="init", steps=4, B=2, n_lo=3, n_hi=9, sanity=False, seed=82),
    "qm9_ddpm60_sanity": dict(net="qm9_ddpm60", regime="init", steps=12, B=3, n_lo=3, n_hi=9, sanity=True, seed=83),
}

# train step of the molecule path (runner/trainer/trainer_node_adj.py:95-126): (case, regime, sigma index, coin)
NODEADJ_TRAIN_CASES = [("qm9_ddpm60", "init", 0, True), ("qm9_onehot96", "stress", 0, False),
                       ("zinc_ddpm96", "stress", 0, True), ("qm9_ddpm60_nosc", "stress", 0, False)]


# attribute codecs (utils/attribute_code.py): number of attribute types k; QM9 has 4 atom / 4 bond types, ZINC 9 / 4
CODEC_CASES = {
    "k4": dict(B=5, N=9, k=4, n_lo=3, n_hi=9, seed=91),
    "k9": dict(B=3, N=20, k=9, n_lo=6, n_hi=20, seed=92),
    "k2": dict(B=4, N=12, k=2, n_lo=2, n_hi=12, seed=93),
}
