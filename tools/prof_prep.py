import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudsegmentation_amd.preprocess import preprocess_point_cloud
rng = np.random.default_rng(0)
n = 166499
pts = (rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64)
full = rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])
col = rng.integers(0, 256, (n, 3)).astype(np.float64)
pg, fg, cg = torch.from_numpy(pts).cuda(), torch.from_numpy(full).cuda(), torch.from_numpy(col).cuda()
for _ in range(3):
    preprocess_point_cloud(pg, cg)
torch.cuda.synchronize()
for _ in range(3):
    preprocess_point_cloud(fg, cg)
torch.cuda.synchronize()
