"""Does the time of an identical sample() call drift over consecutive calls?  And precompute_conditioning on / off
alternating.  grid B = 112, 256 steps."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, swingnn_amd as S
torch.set_grad_enabled(False)
cfg = bench.CONFIGS["grid"]; B = 112; dev = torch.device("cuda:0")
model = bench.build_model(cfg, dev)
flags = bench.make_flags(cfg, B, np.random.default_rng(0)).to(dev)
smp = S.EDMSampler(num_steps=256, clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                   clip_samples_scope="x_0", self_condition=True, dev=dev)
for i, pc in enumerate((False, False, False, True, False, True, False)):
    smp.precompute_conditioning = pc
    np.random.seed(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    smp.sample(model, flags)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    from swingnn_amd import engine as E
    print(i, "precompute" if pc else "default   ", round(dt, 4), "s", round(B / dt, 3), "graphs/s", flush=True)
