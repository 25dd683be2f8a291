import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from src.models.point2net import Point2Net
from pointcloudsegmentation_amd.synth import make_batch, make_labels
from src.models.losses import InstanceSegLoss
torch.manual_seed(0)
m = Point2Net(6, 64).cuda().train()
x = torch.from_numpy(make_batch(4, 2048, seed=3)).cuda()
r = torch.randn(4, 2048, 64, device='cuda')
y = torch.from_numpy(make_labels(4, 2048, seed=1)).cuda()
pts = x[:, :, :3].contiguous()
crit = InstanceSegLoss(64, 0.1, 0.5)
REAL = len(sys.argv) > 1
def run(prec):
    m.precision = prec
    for p in m.parameters(): p.grad = None
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    e = m(x)
    ((crit(pts, e, y)['total_loss']) if REAL else (e * r).sum()).backward()
    g = {k: p.grad.clone() for k, p in m.named_parameters()}
    st = {k: v.clone() for k, v in m.state_dict().items() if 'running' in k}
    m.load_state_dict(sd)
    return e.detach().clone(), g, st
e0, g0, s0 = run("fp32")
e1, g1, s1 = run("bf16")
print("emb rel-to-max", float((e1-e0).abs().max()/e0.abs().max()))
worst = 0
for k in g0:
    sc = float(g0[k].abs().max())
    d = float((g1[k]-g0[k]).abs().max())
    rel = d / max(sc, 1e-30)
    worst = max(worst, rel if sc > 1e-12 else 0)
    cos = float((g1[k]*g0[k]).sum() / (g1[k].norm()*g0[k].norm()).clamp_min(1e-30))
    print("%-40s scale %.3e  rel %.3e  cos %.5f" % (k, sc, rel, cos))
print("worst grad rel", worst)
print("running stats rel", max(float((s1[k]-s0[k]).abs().max()/s0[k].abs().max().clamp_min(1e-12)) for k in s0))
