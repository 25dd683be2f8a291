"""p2n_forward_eval_noalloc (include/p2n.h): the allocation-free variant of p2n_forward_eval.

No counterpart in the reference; it is checked against the existing entry point, through the raw
C ABI (ctypes), bit for bit: both run the same kernels, only the workspace differs (caller buffer vs
the library's built-in device array).  Two calls in a row, the second with a smaller input, because
the array keeps the previous call's contents.  Collected last (file name) so a failure here cannot
hide other tests under `pytest -x`.
"""
import numpy as np
import pytest
import torch

from pointcloudsegmentation_amd import _lib
from pointcloudsegmentation_amd import runtime as rt
from pointcloudsegmentation_amd.synth import make_batch

pytestmark = pytest.mark.gpu


def _packed(ref_params, dev):
    raw = torch.from_numpy(ref_params).to(dev)
    return rt.pack_eval_params(raw, 6, 64)


def _call_both(L, x, packed, want_feats=False):
    B, N, F = x.shape
    dev = x.device
    s = rt.stream_handle(dev)

    def outs():
        emb = torch.full((B, N, 64), float("nan"), device=dev)
        g = torch.full((B, 256), float("nan"), device=dev)
        f = [torch.full((B, c, N), float("nan"), device=dev) for c in (64, 128, 256)] if want_feats else [None] * 3
        return emb, g, f

    emb0, g0, f0 = outs()
    ws = torch.empty(L.p2n_forward_eval_workspace_bytes(B, N, F, 64), dtype=torch.uint8, device=dev)
    _lib.check(L.p2n_forward_eval(x.data_ptr(), B, N, F, 64, packed.data_ptr(), emb0.data_ptr(), g0.data_ptr(),
                                  rt.ptr(f0[0]), rt.ptr(f0[1]), rt.ptr(f0[2]), ws.data_ptr(), ws.numel(), s),
               "p2n_forward_eval")
    emb1, g1, f1 = outs()
    _lib.check(L.p2n_forward_eval_noalloc(x.data_ptr(), B, N, F, 64, packed.data_ptr(), emb1.data_ptr(),
                                          g1.data_ptr(), rt.ptr(f1[0]), rt.ptr(f1[1]), rt.ptr(f1[2]), s),
               "p2n_forward_eval_noalloc")
    torch.cuda.synchronize(dev)
    return (emb0, g0, f0), (emb1, g1, f1)


def test_noalloc_matches_workspace_entry_point_twice(ref_params):
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    packed = _packed(ref_params, dev)
    # first call: the largest shape the built-in scratch takes (256 workgroups of 128 points);
    # second call: smaller and ragged, on the scratch the first one left behind
    for (B, N, seed, feats) in ((4, 8192, 3, False), (3, 1000, 4, True), (1, 1, 5, False)):
        x = torch.from_numpy(make_batch(B, N, seed)).to(dev)
        (emb0, g0, f0), (emb1, g1, f1) = _call_both(L, x, packed, feats)
        assert torch.isfinite(emb1).all()
        assert torch.equal(emb0, emb1), f"emb differs at B={B} N={N}"      # bit-exact: same kernels
        assert torch.equal(g0, g1)
        if feats:
            for a, b in zip(f0, f1):
                assert torch.equal(a, b)


def test_noalloc_allocates_nothing_and_rejects_oversize(ref_params):
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    packed = _packed(ref_params, dev)
    x = torch.from_numpy(make_batch(2, 4096, 6)).to(dev)
    emb = torch.empty((2, 4096, 64), device=dev)
    s = rt.stream_handle(dev)
    L.p2n_forward_eval_noalloc(x.data_ptr(), 2, 4096, 6, 64, packed.data_ptr(), emb.data_ptr(), None, None, None, None, s)
    torch.cuda.synchronize(dev)
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(5):
        assert L.p2n_forward_eval_noalloc(x.data_ptr(), 2, 4096, 6, 64, packed.data_ptr(), emb.data_ptr(), None,
                                          None, None, None, s) == 0
    torch.cuda.synchronize(dev)
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, "device memory in use changed across allocation-free calls"
    # one workgroup beyond the built-in scratch: a clear error, nothing launched
    B, N = 257, 128
    xb = torch.zeros((B, N, 6), device=dev)
    eb = torch.zeros((B, N, 64), device=dev)
    rc = L.p2n_forward_eval_noalloc(xb.data_ptr(), B, N, 6, 64, packed.data_ptr(), eb.data_ptr(), None, None, None,
                                    None, s)
    assert rc == 1
    assert b"exceeds the built-in scratch" in L.p2n_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "p2n_forward_eval_noalloc")
    assert L.p2n_forward_eval_noalloc(xb.data_ptr(), 0, N, 6, 64, packed.data_ptr(), eb.data_ptr(), None, None, None,
                                      None, s) == 0


def test_noalloc_takes_its_scratch_from_the_tensors_device(ref_params):
    """The built-in scratch is per GPU: the entry point looks the device up from x itself and refuses anything that is
    not device memory on the current device (a host pointer here; on a multi-GPU node a tensor of another GPU gets the
    same P2N_EINVAL instead of another GPU's scratch address)."""
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    packed = _packed(ref_params, dev)
    host = torch.zeros((1, 128, 6))                      # pageable host memory: not a device pointer
    emb = torch.empty((1, 128, 64), device=dev)
    rc = L.p2n_forward_eval_noalloc(host.data_ptr(), 1, 128, 6, 64, packed.data_ptr(), emb.data_ptr(), None, None, None,
                                    None, rt.stream_handle(dev))
    assert rc != 0
    msg = L.p2n_last_error()
    assert b"p2n_forward_eval_noalloc" in msg or b"hipPointerGetAttributes" in msg
    # and the call still works afterwards
    x = torch.from_numpy(make_batch(1, 128, 3)).to(dev)
    assert L.p2n_forward_eval_noalloc(x.data_ptr(), 1, 128, 6, 64, packed.data_ptr(), emb.data_ptr(), None, None, None,
                                      None, rt.stream_handle(dev)) == 0
    torch.cuda.synchronize(dev)
    assert bool(torch.isfinite(emb).all())
