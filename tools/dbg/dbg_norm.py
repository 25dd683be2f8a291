import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from pointcloudsegmentation_amd.clustering import normalize_rows
for D in [1, 2, 5, 8, 64, 129, 255]:
    e = np.random.default_rng(D).standard_normal((1000, D)).astype(np.float32) * 3
    e[5] = 0.0
    want = e / (np.linalg.norm(e, axis=1, keepdims=True) + 1e-8)
    got = normalize_rows(e)
    bad = np.argwhere(got != want)
    print(D, len(bad), flush=True)
    for r, c in bad[:4]:
        nrm = np.linalg.norm(e[r:r+1], axis=1)[0]
        print("   ", r, c, repr(e[r, c]), repr(got[r, c]), repr(want[r, c]), repr(nrm), repr(np.float32(nrm) + np.float32(1e-8)))
