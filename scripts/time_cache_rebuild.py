"""Host/GPU time of the derived-weight cache rebuild that follows every optimizer step (engine.NetCache + train.TrainCache)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swingnn_amd as S
from swingnn_amd import engine as E, train as T
dev = "cuda:0"
net = S.SwinGNN(img_size=24, patch_size=3, in_chans=1, embed_dim=96, depths=[4, 4, 6], num_heads=[3, 6, 12], window_size=24,
                mlp_ratio=4., drop_rate=0., attn_drop_rate=0., drop_path_rate=0., self_condition=True).to(dev)
for it in range(4):
    with torch.no_grad():
        torch._foreach_add_([p.view(-1)[:0] for p in net.parameters()], 0.0)
    torch.cuda.synchronize(); t0 = time.time()
    c = E.get_cache(net)
    t1 = time.time()
    tc = T.get_train_cache(net, c)
    t2 = time.time(); torch.cuda.synchronize(); t3 = time.time()
    print(f"NetCache host {1e3*(t1-t0):.2f} ms, TrainCache host {1e3*(t2-t1):.2f} ms, total incl. GPU drain {1e3*(t3-t0):.2f} ms")
