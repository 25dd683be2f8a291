import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudsegmentation_amd.synth import make_batch
from src.models.point2net import PointNet2Encoder

if len(sys.argv) > 1 and sys.argv[1] == "scan":      # whole-model input-gradient case of tests/test_train_gpu.py per seed
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_train_gpu as T
    for seed in range(4, 16):
        print("seed %d: emb %.2e  d/d points %.2e  worst parameter gradient %.2e" % ((seed,) + T._input_gradient_case(seed)), flush=True)
    sys.exit(0)


def rel(a, b):
    a = a.detach().double().cpu().numpy(); b = b.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))

SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(SEED)
ref32 = PointNet2Encoder(6).train()
gpu = copy.deepcopy(ref32).cuda()
ref = copy.deepcopy(ref32).double()
x = torch.from_numpy(make_batch(3, 700, seed=8))
w = [torch.randn(3, c, 700) for c in (64, 128, 256)] + [torch.randn(3, 256)]
use_global = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def run(m, x, cast):
    outs = m(x)
    for o in outs[:3]:
        o.retain_grad()
    terms = [(o * cast(wi)).sum() for o, wi in zip(outs, w)]
    if not use_global:
        terms = terms[:3]
    sum(terms).backward()
    return outs
o64 = run(ref, x.double().requires_grad_(True), lambda t: t.double())
o32 = run(ref32, x.clone().requires_grad_(True), lambda t: t)
og = run(gpu, x.cuda().requires_grad_(True), lambda t: t.cuda())
for i, n in enumerate(("f1", "f2", "f3")):
    print(n, "value GPU %.2e CPU32 %.2e | grad GPU %.2e CPU32 %.2e" % (rel(og[i], o64[i]), rel(o32[i], o64[i]), rel(og[i].grad, o64[i].grad), rel(o32[i].grad, o64[i].grad)))
print("global value GPU %.2e" % rel(og[3], o64[3]))
# which elements differ in f2.grad
d = (og[1].grad.double().cpu() - o64[1].grad).abs()
print("f2.grad: max diff", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape), "ref max", float(o64[1].grad.abs().max()), "n>1e-4*max:", int((d > 1e-4 * o64[1].grad.abs().max()).sum()), "of", d.numel())
d = (og[2].grad.double().cpu() - o64[2].grad).abs()
print("f3.grad: max diff", float(d.max()), "n>1e-6*max:", int((d > 1e-6 * o64[2].grad.abs().max()).sum()))
