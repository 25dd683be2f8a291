#!/usr/bin/env python3
"""Where does the training-step gradient deviate at full size?  GPU vs float64 CPU module, per tensor and
for embedding_net.0.weight per half (f3 columns / global-feature columns)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudsegmentation_amd.synth import make_batch
from src.models.point2net import Point2Net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(17)
mg = Point2Net(6, 64)
m64 = Point2Net(6, 64); m64.load_state_dict(mg.state_dict()); m64 = m64.double().train()
m32 = Point2Net(6, 64); m32.load_state_dict(mg.state_dict()); m32.train()
mg = mg.cuda().train()
x = torch.from_numpy(make_batch(B, 8192, seed=31)); w = torch.randn(B, 8192, 64)
(m64(x.double()) * w.double()).sum().backward()
(m32(x) * w).sum().backward()
(mg(x.cuda()) * w.cuda()).sum().backward()
for (k, p64), (_, p32), (_, pg) in zip(m64.named_parameters(), m32.named_parameters(), mg.named_parameters()):
    sc = max(p64.grad.abs().max().item(), 1e-3)
    eg = (pg.grad.cpu().double() - p64.grad).abs().max().item() / sc
    e32 = (p32.grad.double() - p64.grad).abs().max().item() / sc
    print(f"{k:32s} scale {sc:9.3e} gpu {eg:8.2e} cpu32 {e32:8.2e}")
    if k == "embedding_net.0.weight":
        g64 = p64.grad[:, :, 0]; gg = pg.grad.cpu().double()[:, :, 0]; g32 = p32.grad.double()[:, :, 0]
        for name, sl in (("f3 half", slice(0, 256)), ("global half", slice(256, 512))):
            print(f"   {name}: max|ref| {g64[:, sl].abs().max():.3e} gpu abs err {(gg[:, sl]-g64[:, sl]).abs().max():.3e} cpu32 abs err {(g32[:, sl]-g64[:, sl]).abs().max():.3e}")
