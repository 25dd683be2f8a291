"""Not a pytest file: first-light script for the GPU box (parity + timing of the eval forward)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudsegmentation_amd.synth import make_batch
from src.models.point2net import Point2Net
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sd = dict(np.load(os.path.join(G, "weights.npz")))
m = Point2Net(6, 64)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda().eval()
for name in ["eval_b1_n7_s1", "eval_b2_n1000_s0", "eval_b3_n4099_s1"]:
    g = np.load(os.path.join(G, name + ".npz"))
    with torch.no_grad():
        x = torch.from_numpy(g["x"]).cuda()
        emb = model_out = m(x)
        f1, f2, f3, gf = m.encoder(x)
    torch.cuda.synchronize()
    e = np.abs(emb.cpu().numpy() - g["emb"]).max() / np.abs(g["emb"]).max()
    eg = np.abs(gf.cpu().numpy() - g["global_f"]).max() / np.abs(g["global_f"]).max()
    print(name, "emb relmax err", e, "global_f", eg, flush=True)
    if "f3" in g.files:
        for k, t in (("f1", f1), ("f2", f2), ("f3", f3)):
            print("  ", k, np.abs(t.cpu().numpy() - g[k]).max() / np.abs(g[k]).max(), flush=True)
x = torch.from_numpy(make_batch(16, 8192, 0)).cuda()
with torch.no_grad():
    for _ in range(3): y = m(x)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20): y = m(x)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 20
print(f"B=16 N=8192: {dt*1e3:.3f} ms/fwd  {16*8192/dt/1e6:.1f} Mpts/s  {16*8192*467712/dt/1e12:.1f} TFLOP/s (fold form)")
