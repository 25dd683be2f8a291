import sys, ctypes; sys.path.insert(0, '/root/repo')
import torch
from pointcloudsegmentation_amd import ops, _lib
L = _lib.lib()
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize(); L.p2n_profile_enable(1)
    for _ in range(iters): fn()
    ms = (ctypes.c_float * 32)(); cnt = (ctypes.c_int * 32)(); L.p2n_profile_collect(32, ms, cnt); L.p2n_profile_enable(0)
    return sum(ms) / iters
for B in (16, 64, 256):
    C, N, M, ns = 64, 8192, 2048, 32
    feat = torch.randn(B, C, N, device='cuda'); idx = torch.randint(0, N, (B, M, ns), device='cuda', dtype=torch.int32)
    ms = timed(lambda: ops.group_gather(feat, idx))
    by = B * (4*C*M*ns + 4*M*ns + 4*C*N)
    print('gather B', B, ms, by/ms/1e6, 'GB/s')
    w = torch.rand(B, N, 3, device='cuda'); i3 = torch.randint(0, M, (B, N, 3), device='cuda', dtype=torch.int32); f2 = torch.randn(B, 128, M, device='cuda')
    ms = timed(lambda: ops.three_interpolate(f2, i3, w))
    by = B * (4*128*N + 24*N + 4*128*M)
    print('interp B', B, ms, by/ms/1e6, 'GB/s')
x = torch.empty(1<<28, device='cuda'); y = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
y.copy_(x); torch.cuda.synchronize(); e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize(); print('copy 1 GiB r+w GB/s', 10*2*x.numel()*4/e0.elapsed_time(e1)/1e6)
e0.record()
for _ in range(10): y.fill_(1.0)
e1.record(); torch.cuda.synchronize(); print('fill 1 GiB write GB/s', 10*x.numel()*4/e0.elapsed_time(e1)/1e6)
