"""GPU: training the hierarchical sibling (set abstraction / feature propagation) on the device --
p2n_mlp_train_forward / p2n_mlp_train_backward (train-mode shared MLP) plus the operators' adjoints -- against
the same modules on CPU tensors in FLOAT64 (stock PyTorch autograd; indices and interpolation weights from the
float32 coordinates, as on the GPU), i.e. a reference whose own rounding noise is negligible.  Parity unpinned by
the reference (it has no such levels).  Bars: outputs 1e-5 of max, running statistics 1e-5, gradients 5e-4 of
each tensor's max (float32 BatchNorm backward sums cancel heavily; measured values are printed)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(got, want):
    want = want.detach().double().cpu().numpy()
    got = got.detach().double().cpu().numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    # floor: the bias of a convolution in front of a BatchNorm has a zero gradient (exactly 0 on the GPU, 1e-17 in float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.rand(mod.bias.shape, generator=g) * 0.4 - 0.2)
                mod.running_mean.copy_(torch.rand(mod.running_mean.shape, generator=g) * 0.4 - 0.2)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)


def _compare_modules(gpu, ref, what, grad_bar=5e-4):
    worst = 0.0
    for (k, p), (_, q) in zip(gpu.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, k
        e = _rel(p.grad, q.grad)
        worst = max(worst, e)
        assert e <= grad_bar, f"{what}: grad {k}: {e:.3e}"
    for (k, b), (_, c) in zip(gpu.named_buffers(), ref.named_buffers()):
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(c), k
        else:
            assert _rel(b, c) <= 1e-5, f"{what}: buffer {k}: {_rel(b, c):.3e}"
    return worst


@pytest.mark.parametrize("cin,chans,last_plain,B,L,need_dx", [
    (6, [32, 32, 64], False, 2, 4096, False),        # sa1's chain, no input gradient
    (67, [64, 64, 128], False, 3, 1000, True),       # P = 3000: not a multiple of 128, odd input channels
    (384, [256, 256], False, 2, 777, True),          # fp3: more than 256 input channels (two row blocks of the data gradient)
    (128, [128, 64], True, 2, 2048, True),           # the embedding head: plain last layer
    (259, [100, 33], True, 1, 515, True),            # channel counts that fill no tile
    (131, [128], False, 2, 300, True),               # one layer
    (40, [7], True, 2, 300, True),                   # one plain layer
    (8, [24, 40, 72], False, 2, 300, True),          # reduction lengths 8, 24, 40: multiples of 8 with a ragged last k-tile (scalar-base addressing, clamped rows)
    (12, [20, 44, 16], True, 2, 515, True),          # reduction lengths 12, 20, 44: multiples of 4 only (general addressing path)
])
def test_shared_mlp_train_matches_float64_torch(cin, chans, last_plain, B, L, need_dx):
    from pointcloudsegmentation_amd.pointnet2 import SharedMLP
    torch.manual_seed(cin + L)
    ref = SharedMLP([cin] + chans, last_plain=last_plain).train()
    _randomise(ref, cin)
    gpu = copy.deepcopy(ref).cuda()
    ref = ref.double()
    x = torch.randn(B, cin, L) * 0.7 + 0.1
    r = torch.randn(B, chans[-1], L)
    xr = x.double().requires_grad_(need_dx)
    yr = ref(xr)
    (yr * r.double()).sum().backward()
    xg = x.cuda().requires_grad_(need_dx)
    yg = gpu(xg)
    (yg * r.cuda()).sum().backward()
    assert yg.shape == yr.shape
    e_out = _rel(yg, yr)
    assert e_out <= 1e-5, f"output {e_out:.3e}"
    worst = _compare_modules(gpu, ref, "shared MLP")
    e_dx = 0.0
    if need_dx:
        e_dx = _rel(xg.grad, xr.grad)
        assert e_dx <= 5e-4, f"dx {e_dx:.3e}"
    print(f"shared MLP {cin}->{chans}: out {e_out:.2e}, worst grad {worst:.2e}, dx {e_dx:.2e}")


def test_shared_mlp_train_under_no_grad_and_second_backward():
    from pointcloudsegmentation_amd.pointnet2 import SharedMLP
    torch.manual_seed(3)
    ref = SharedMLP([16, 32, 32]).train()
    gpu = copy.deepcopy(ref).cuda()
    x = torch.randn(2, 16, 640)
    with torch.no_grad():
        want = ref(x)
        got = gpu(x.cuda())
    assert _rel(got, want) <= 1e-5
    assert int(gpu.net[1].num_batches_tracked) == 1
    assert _rel(gpu.net[1].running_var, ref.net[1].running_var) <= 1e-5
    y = gpu(x.cuda())
    y.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError):
        y.sum().backward()


@pytest.mark.parametrize("training", [True, False])
def test_set_abstraction_and_propagation_train(training):
    """training=False: eval() with autograd enabled -- the levels differentiate through BatchNorm on its running
    statistics (which stay as they are), against stock PyTorch in float64."""
    from pointcloudsegmentation_amd.pointnet2 import FeaturePropagation, SetAbstraction
    from pointcloudsegmentation_amd.synth import make_batch
    torch.manual_seed(5)
    x = torch.from_numpy(make_batch(2, 1024, seed=22))
    xyz, feat = x[:, :, :3].contiguous(), x[:, :, 3:].transpose(1, 2).contiguous()
    sa = SetAbstraction(4, 0.15, 16, 3, [16, 32]).train(training)
    fp = FeaturePropagation(32 + 3, [64, 48]).train(training)
    _randomise(sa, 1); _randomise(fp, 2)
    sa_g, fp_g = copy.deepcopy(sa).cuda(), copy.deepcopy(fp).cuda()
    sa, fp = sa.double(), fp.double()
    r = torch.randn(2, 48, 1024)

    f64 = feat.double().requires_grad_(True)
    nx, nf = sa(xyz.double(), f64)
    up = fp(xyz.double(), nx, f64, nf)
    (up * r.double()).sum().backward()

    fg = feat.cuda().requires_grad_(True)
    nxg, nfg = sa_g(xyz.cuda(), fg)
    upg = fp_g(xyz.cuda(), nxg, fg, nfg)
    (upg * r.cuda()).sum().backward()

    np.testing.assert_array_equal(nxg.cpu().numpy(), nx.float().numpy())
    assert _rel(nfg, nf) <= 1e-5 and _rel(upg, up) <= 1e-5
    w1 = _compare_modules(sa_g, sa, "SA")
    w2 = _compare_modules(fp_g, fp, "FP")
    e = _rel(fg.grad, f64.grad)
    assert e <= 5e-4, f"d feat {e:.3e}"
    print(f"SA/FP train: worst grad SA {w1:.2e} FP {w2:.2e}, d feat {e:.2e}")


def test_point2net_sa_training_step_on_gpu():
    """The whole sibling, forward + backward.  The gradient of the first levels has passed seven train-mode
    BatchNorm chains whose float32 backward sums cancel heavily, so the bar per tensor is the noise floor of
    float32 itself: max(5e-4, 4 x the deviation of the SAME module run by stock PyTorch in float32 on the CPU
    from its float64 run)."""
    from pointcloudsegmentation_amd.pointnet2 import Point2NetSA
    from pointcloudsegmentation_amd.synth import make_batch
    torch.manual_seed(7)
    ref32 = Point2NetSA(6, 64).train()
    _randomise(ref32, 9)
    gpu = copy.deepcopy(ref32).cuda()
    ref = copy.deepcopy(ref32).double()
    x = torch.from_numpy(make_batch(2, 2048, seed=35))
    r = torch.randn(2, 2048, 64)
    yr = ref(x.double())
    (yr * r.double()).sum().backward()
    y32 = ref32(x)
    (y32 * r).sum().backward()
    yg = gpu(x.cuda())
    (yg * r.cuda()).sum().backward()
    assert yg.shape == (2, 2048, 64)
    e_out = _rel(yg, yr)
    assert e_out <= 2e-5, f"embeddings {e_out:.3e} (CPU float32: {_rel(y32, yr):.3e})"
    worst = worst_noise = 0.0
    for (k, p), (_, q), (_, c) in zip(gpu.named_parameters(), ref.named_parameters(), ref32.named_parameters()):
        e, noise = _rel(p.grad, q.grad), _rel(c.grad, q.grad)
        print(f"  {k:28s} |ref|max {float(q.grad.abs().max()):.3e}  GPU {e:.2e}  CPU float32 {noise:.2e}")
        if float(q.grad.abs().max()) > 1e-6:        # (biases in front of a BatchNorm: true gradient 0, exactly 0 on the GPU)
            worst, worst_noise = max(worst, e), max(worst_noise, noise)
        assert e <= max(5e-4, 4 * noise), f"grad {k}: GPU {e:.3e}, CPU float32 {noise:.3e}"
    for (k, b), (_, c) in zip(gpu.named_buffers(), ref.named_buffers()):
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(c), k
        else:
            assert _rel(b, c) <= 1e-5, f"buffer {k}: {_rel(b, c):.3e}"
    print(f"Point2NetSA train 2x2048: out {e_out:.2e}, worst grad GPU {worst:.2e} / CPU float32 {worst_noise:.2e}")
    # one optimiser step on the device, then eval: the fused level images must follow the new weights
    opt = torch.optim.SGD(gpu.parameters(), lr=1e-3)
    opt.step()
    with torch.no_grad():
        for p, q in zip(gpu.parameters(), ref.parameters()):
            q.copy_(p.double().cpu())
        for b, c in zip(gpu.buffers(), ref.buffers()):
            c.copy_(b.cpu())
        want = ref.eval()(x.double())
        got = gpu.eval()(x.cuda())
    assert _rel(got, want) <= 2e-5


def test_eval_images_follow_running_statistics_updated_without_an_optimizer_step():
    """train-mode forwards under no_grad change only the BatchNorm buffers, through raw pointers: the packed eval
    images (fused levels of Point2NetSA; Point2Net's image when its encoder was run on its own) must not go stale"""
    from pointcloudsegmentation_amd.pointnet2 import Point2NetSA
    from pointcloudsegmentation_amd.synth import make_batch
    from src.models.point2net import Point2Net
    torch.manual_seed(12)
    x = torch.from_numpy(make_batch(2, 1024, seed=5))
    for make in (lambda: Point2NetSA(6, 64), lambda: Point2Net(6, 64)):
        ref = make()
        gpu = copy.deepcopy(ref).cuda()
        with torch.no_grad():
            a = gpu.eval()(x.cuda())                      # builds the eval image(s)
            ref.eval()(x)
            gpu.train(); ref.train()
            if isinstance(ref, Point2Net):                # the bare encoder, trained on its own
                gpu.encoder(x.cuda()); ref.encoder(x)
            else:
                gpu(x.cuda()); ref(x)
            want = ref.eval()(x)
            got = gpu.eval()(x.cuda())
        assert _rel(got, want) <= 2e-5, f"{type(ref).__name__}: stale eval image ({_rel(got, want):.2e})"
        assert _rel(got, a) > 1e-3                       # the statistics did move


@pytest.mark.parametrize("C,G,ns,pad", [(3, 10, 8, 0), (64, 2048, 32, 0), (5, 37, 16, 40), (2, 9, 64, 64), (7, 50, 5, 2), (1, 1, 1, 3)])
def test_group_max_on_channel_major_rows_and_its_backward(C, G, ns, pad):
    """p2n_group_max_cm / p2n_group_max_cm_bwd (the set-abstraction levels' max over a group in training, taken on the
    channel-major output of the train-mode shared MLP) against torch: values and first arg-max positions bit-exact on
    rows with negative values, exact ties and an all-equal group; the backward equals autograd's routing and zeroes
    the pad columns."""
    from pointcloudsegmentation_amd import _lib, runtime as _rt
    L = _lib.lib()
    ld = (G * ns + pad + 3) // 4 * 4
    g = torch.Generator().manual_seed(C * 131 + ns)
    x = torch.randn(C, ld, generator=g)
    x[:, :G * ns].view(C, G, ns)[:, ::3, :] = x[:, :G * ns].view(C, G, ns)[:, ::3, :].round()      # exact ties
    x[0, :ns] = 0.25                                                                                # an all-equal group
    xg = x.cuda()
    out = torch.empty(C, G, device="cuda")
    arg = torch.empty(C, G, dtype=torch.uint8, device="cuda")
    _lib.check(L.p2n_group_max_cm(xg.data_ptr(), C, ld, G, ns, out.data_ptr(), arg.data_ptr(), _rt.stream_handle(xg.device)),
               "p2n_group_max_cm")
    v = x[:, :G * ns].view(C, G, ns)
    want = v.max(dim=2).values
    first = (v == want.unsqueeze(2)).to(torch.uint8).argmax(dim=2)
    assert torch.equal(out.cpu(), want)
    assert torch.equal(arg.cpu().long(), first)
    gout = torch.randn(C, G, generator=g)
    din = torch.full((C, ld), float("nan"), device="cuda")
    _lib.check(L.p2n_group_max_cm_bwd(gout.cuda().data_ptr(), arg.data_ptr(), C, G, ns, ld, din.data_ptr(),
                                      _rt.stream_handle(xg.device)), "p2n_group_max_cm_bwd")
    ref = torch.zeros(C, ld)
    ref[:, :G * ns].view(C, G, ns).scatter_(2, first.unsqueeze(2), gout.unsqueeze(2))
    assert torch.equal(din.cpu(), ref)
