#!/usr/bin/env python
"""CPU model of the multi-winner FPS round (csrc/p2n_ops.hip: p2n_fps_rounds_kernel): how many samples one exchange
round accepts, and that the accepted sequence IS the sequential FPS sequence.  numpy float32 arithmetic with the
kernel's operation order."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudsegmentation_amd.synth import make_batch


def sqd(P, c):
    d = P - c
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def fps_plain(P, M, start):
    md = np.full(len(P), np.inf, np.float32); out = [start]; cur = start
    for _ in range(M - 1):
        md = np.minimum(md, sqd(P, P[cur])); cur = int(np.argmax(md)); out.append(cur)
    return out


def fps_rounds(P, M, start, G, top=1):
    N = len(P); chunk = (N + G - 1) // G
    md = np.full(N, np.inf, np.float32); out = [start]
    md = np.minimum(md, sqd(P, P[start]))
    rounds = 0; hist = []
    bounds = [(g * chunk, min(N, (g + 1) * chunk)) for g in range(G)]
    while len(out) < M:
        rounds += 1
        k1 = []; d2 = []
        for lo, hi in bounds:
            if hi <= lo: continue
            seg = md[lo:hi]; i = int(np.argmax(seg)); k1.append((seg[i], -(lo + i)))   # key order: dist, then lowest index
            s = np.partition(seg, -2)[-2] if hi - lo > 1 else np.float32(0); d2.append(s)
        S = max(d2)
        order = sorted(range(len(k1)), key=lambda j: k1[j], reverse=True)
        acc = []
        for r, j in enumerate(order):
            d, ni = k1[j]; i = -ni
            if r > 0 and not d > S: break
            if any(sqd(P[i:i + 1], P[a])[0] < d or not np.isfinite(P[a].sum()) for a in acc): break
            acc.append(i)
            if len(out) + len(acc) >= M: break
        for a in acc:
            md = np.minimum(md, sqd(P, P[a]))
        out += acc; hist.append(len(acc))
    return out, rounds, hist


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    P = np.ascontiguousarray(make_batch(1, N, 3)[0, :, :3]).astype(np.float32)
    ref = fps_plain(P, M, 0)
    for G in [int(a) for a in sys.argv[3:]] or [8, 16, 32, 64]:
        got, rounds, hist = fps_rounds(P, M, 0, G)
        h = np.array(hist)
        print(f"N={N} M={M} G={G}: identical={got == ref} rounds={rounds} samples/round={M / rounds:.2f} "
              f"first100={h[:100].mean():.2f} last100={h[-100:].mean():.2f} max={h.max()}")
