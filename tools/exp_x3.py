#!/usr/bin/env python3
"""fp32x3 eval path at the headline shape: deviation from the exact-fp32 path and from the oracle-pinned golden,
kernel times (HIP events) for both, and what a NaN/Inf input row does on the exact-fp32 path."""
import ctypes, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
from pointcloudsegmentation_amd import _lib
from pointcloudsegmentation_amd.synth import make_batch

dev = torch.device("cuda", 0)
lib = _lib.lib()
m = bench.load_model(dev)
x = torch.from_numpy(make_batch(16, 8192, 1000)).to(dev)
out = {}
with torch.no_grad():
    for prec in ("fp32", "fp32x3", "bf16"):
        m.precision = prec
        for _ in range(30):
            y = m(x)
        torch.cuda.synchronize()
        lib.p2n_profile_select(-1); lib.p2n_profile_enable(1)
        for _ in range(20):
            y = m(x)
        torch.cuda.synchronize()
        k = bench.collect_kernel_ms(); lib.p2n_profile_enable(0)
        t0 = time.perf_counter()
        for _ in range(100):
            y = m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        out[prec] = {"ms_per_step": 1e3 * dt, "Mpts_per_s": 16 * 8192 / dt / 1e6, "kernels_ms": {n: v[0] for n, v in k.items() if v[1]}}
        if prec == "fp32":
            ref = y.clone()
        else:
            out[prec]["max_dev_vs_fp32_path"] = float((y - ref).abs().max() / ref.abs().max())
    m.precision = "fp32"
    # non-finite input rows
    xb = torch.from_numpy(make_batch(3, 300, 5)).to(dev)
    clean = m(xb).clone()
    gclean = m.encoder(xb)[3].clone()
    probe = {}
    for name, val in (("nan", float("nan")), ("-nan", -float("nan")), ("inf", float("inf")), ("-inf", -float("inf"))):
        xq = xb.clone()
        xq[1, 17, 0] = val
        e = m(xq); g = m.encoder(xq)[3]
        probe[name] = {"other_clouds_identical": bool(torch.equal(e[0], clean[0]) and torch.equal(e[2], clean[2])),
                       "other_points_of_cloud_finite": bool(torch.isfinite(torch.cat([e[1, :17], e[1, 18:]])).all()),
                       "bad_point_row_nan_count": int(torch.isnan(e[1, 17]).sum()), "bad_point_row_inf_count": int(torch.isinf(e[1, 17]).sum()),
                       "global_f_nan": int(torch.isnan(g[1]).sum()), "global_f_inf": int(torch.isinf(g[1]).sum()),
                       "global_f_other_identical": bool(torch.equal(g[0], gclean[0]) and torch.equal(g[2], gclean[2])),
                       "other_points_same_as_clean": bool(torch.equal(torch.cat([e[1, :17], e[1, 18:]]), torch.cat([clean[1, :17], clean[1, 18:]])))}
    out["nonfinite_probe_fp32_path"] = probe
print(json.dumps(out, indent=1))
