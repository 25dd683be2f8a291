#!/usr/bin/env python3
"""HBM rate probes on the box (torch fill / copy / read-reduce), to put the write-dominated gather
kernels next to what the memory system does for pure writes and pure copies."""
import json, torch
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
res = {}
for mb in (64, 268, 1024):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    res[f"fill_{mb}MB_GBps"] = round(4 * n / t(lambda: x.fill_(1.0)) / 1e6)
    res[f"copy_{mb}MB_GBps_rw"] = round(8 * n / t(lambda: y.copy_(x)) / 1e6)
    res[f"sum_{mb}MB_GBps"] = round(4 * n / t(lambda: x.sum()) / 1e6)
print(json.dumps(res))
