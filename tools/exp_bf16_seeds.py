import sys, torch
sys.path.insert(0,'/root/repo')
from src.models.point2net import Point2Net
from pointcloudsegmentation_amd.synth import make_batch
def dev(m, x):
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    out = []
    for prec in ("fp32", "bf16"):
        m.load_state_dict(sd); m.precision = prec
        with torch.no_grad(): out.append(m(x).clone())
    m.load_state_dict(sd); m.precision = "fp32"
    return float((out[1]-out[0]).abs().max()/out[0].abs().max())
for seed in range(6):
    torch.manual_seed(seed)
    m = Point2Net(6, 64).cuda().train()
    x = torch.from_numpy(make_batch(4, 2048, seed=10+seed)).cuda()
    print("random init seed", seed, "4x2048:", round(dev(m, x), 5))
torch.manual_seed(0)
m = Point2Net(6, 64).cuda().train()
x = torch.from_numpy(make_batch(32, 8192, seed=200)).cuda()
print("random init 32x8192:", round(dev(m, x), 5))
import os
ck = "/root/repo/tests/golden/ref_weights.pt"
print(os.path.exists(ck))
