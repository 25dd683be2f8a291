#!/usr/bin/env python3
"""FPS timing sweep over the workgroup width (P2N_FPS_T) and workgroups per cloud (P2N_FPS_G)."""
import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = os.path.join(REPO, "tools", "fps_sweep.py")
for (B, N, M, combos) in [(16, 8192, 2048, [(0, 0), (512, 0)]), (16, 2048, 512, [(0, 0), (512, 0)]), (16, 512, 128, [(0, 0), (512, 0)]),
                          (1, 40000, 10000, [(0, 0), (512, 0), (512, 8), (512, 5), (512, 4), (0, 5), (0, 4)]),
                          (8, 40000, 2000, [(0, 0), (512, 0), (512, 4)])]:
    for (t, g) in combos:
        env = dict(os.environ)
        env.pop("P2N_FPS_G", None); env.pop("P2N_FPS_T", None)
        if t: env["P2N_FPS_T"] = str(t)
        if g: env["P2N_FPS_G"] = str(g)
        r = subprocess.run([sys.executable, child, "child", str(B), str(N), str(M)], env=env, capture_output=True, text=True, timeout=300)
        print(f"T={t or 256} G={g or 'auto'}", r.stdout.strip() or r.stderr[-300:], flush=True)
