"""Host-side cost of a Heun step, eager launches vs EDMSampler.use_graphs: wall time between the starts of consecutive
steps as seen by the host thread while the GPU queue is still behind (the loop never syncs), and the end-to-end rate.
    python scripts/exp_graph_host_time.py [config] [batch] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import swingnn_amd as S  # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "grid"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 112
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[config]
    model = bench.build_model(cfg, dev)
    flags = bench.make_flags(cfg, B, np.random.default_rng(0)).to(dev)

    class Timed(S.EDMSampler):
        def _randn_like(self, x):
            self.marks.append(time.perf_counter())
            return torch.randn_like(x)

    for graphs in (0, 1):
        smp = Timed(num_steps=steps, clip_samples=False, clip_samples_min=-1.0, clip_samples_max=1.0,
                    clip_samples_scope="x_0", self_condition=True, dev=dev)
        smp.use_graphs = bool(graphs)
        with torch.no_grad():
            for rep in range(3):
                smp.marks = []
                np.random.seed(3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                smp.sample(model, flags)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        d = np.diff(np.array(smp.marks))
        print(f"{config} B={B} {'graphs' if graphs else 'eager '}: host {1e3 * np.median(d):7.3f} ms per Heun step (median of "
              f"{len(d)}), wall {1e3 * dt / steps:7.3f} ms per step")


if __name__ == "__main__":
    main()
