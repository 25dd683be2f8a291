#!/usr/bin/env python3
"""FPS: per-iteration exchange through one XCD's L2 (default where it fits) vs the write-through form (P2N_FPS_NOXCD=1)."""
import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = os.path.join(REPO, "tools", "fps_sweep.py")
for (B, N, M) in [(1, 40000, 10000), (8, 40000, 2000), (16, 20000, 2000), (1, 150000, 4000), (4, 100000, 2000)]:
    for nox in (0, 1):
        for g in (0, 8, 16) if B == 1 and N == 40000 else (0,):
            env = dict(os.environ)
            for k in ("P2N_FPS_G", "P2N_FPS_T", "P2N_FPS_NOXCD"):
                env.pop(k, None)
            if nox: env["P2N_FPS_NOXCD"] = "1"
            if g: env["P2N_FPS_G"] = str(g)
            r = subprocess.run([sys.executable, child, "child", str(B), str(N), str(M)], env=env, capture_output=True, text=True, timeout=300)
            print(f"noxcd={nox} G={g or 'auto'}", r.stdout.strip() or r.stderr[-300:], flush=True)
