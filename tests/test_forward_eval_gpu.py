"""GPU parity of the fused eval forward (p2n_forward_eval) against
  (1) the golden vectors produced by the reference itself, and
  (2) the CPU oracle on fresh seeded inputs,
through the host module (which calls the C ABI).  Tolerance: 1e-5 relative to
the tensor's max magnitude (BASELINE.json north_star: 1e-5 relative in fp32)."""
import numpy as np
import pytest
import torch

from conftest import assert_close_to_max, golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _model(sd_np, embedding_dim=64):
    from src.models.point2net import Point2Net
    m = Point2Net(feature_dim=6, embedding_dim=embedding_dim)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    return m.cuda().eval()


@pytest.fixture(scope="module")
def model(ref_weights):
    return _model(ref_weights)


@pytest.mark.parametrize("name", ["eval_b1_n1_s0", "eval_b1_n7_s1", "eval_b2_n1000_s0", "eval_b3_n4099_s1"])
def test_golden_embeddings(model, name):
    g = golden(name + ".npz")
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        emb = model(x)
        f1, f2, f3, gf = model.encoder(x)
    assert emb.shape == g["emb"].shape and emb.dtype == torch.float32
    assert_close_to_max(emb.cpu().numpy(), g["emb"], TOL, "emb")
    assert_close_to_max(gf.cpu().numpy(), g["global_f"], TOL, "global_f")
    if "f3" in g.files:
        assert_close_to_max(f1.cpu().numpy(), g["f1"], TOL, "f1")
        assert_close_to_max(f2.cpu().numpy(), g["f2"], TOL, "f2")
        assert_close_to_max(f3.cpu().numpy(), g["f3"], TOL, "f3")


def test_golden_headline_shape(model):
    """B=16, N=8192 (BASELINE.json configs[1]); inputs regenerated from the seed."""
    from pointcloudsegmentation_amd.synth import make_batch
    g = golden("eval_b16_n8192_s0.npz")
    x = make_batch(16, 8192, int(g["seed"]))
    np.testing.assert_array_equal(x[:, :4, :], g["x_head"])
    with torch.no_grad():
        emb = model(torch.from_numpy(x).cuda())
        gf = model.encoder(torch.from_numpy(x).cuda())[3]
    assert_close_to_max(emb.cpu().numpy()[:, ::64, :], g["emb_stride64"], TOL, "emb")
    assert_close_to_max(gf.cpu().numpy(), g["global_f"], TOL, "global_f")


def test_golden_other_embedding_dim():
    g = golden("eval_e32_b2_n333_s5.npz")
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    m = _model(sd, embedding_dim=32)
    with torch.no_grad():
        emb = m(torch.from_numpy(g["x"]).cuda())
    assert_close_to_max(emb.cpu().numpy(), g["emb"], TOL, "emb E=32")


@pytest.mark.parametrize("B,N", [(1, 31), (1, 32), (1, 33), (2, 127), (2, 128), (2, 129), (5, 640), (1, 40000),
                                 (1, 150000), (1, 166499)])   # BASELINE configs[2] / configs[0] full scenes
def test_vs_oracle_ragged_sizes(model, ref_params, B, N):
    from oracle import oracle
    from pointcloudsegmentation_amd.synth import make_batch
    x = make_batch(B, N, seed=100 + N)
    want = oracle.forward_eval(x, ref_params, want_features=(N <= 640))
    with torch.no_grad():
        emb = model(torch.from_numpy(x).cuda())
        f1, f2, f3, gf = model.encoder(torch.from_numpy(x).cuda())
    assert_close_to_max(emb.cpu().numpy(), want["emb"], TOL, "emb")
    assert_close_to_max(gf.cpu().numpy(), want["global_f"], TOL, "global_f")
    if N <= 640:
        assert_close_to_max(f3.cpu().numpy(), want["f3"], TOL, "f3")


@pytest.mark.parametrize("E", [1, 3, 48, 100, 256])
def test_vs_oracle_embedding_dims(E):
    """Random weights, embedding_dim not a multiple of 32 / of 4 (scalar store path)."""
    from oracle import oracle
    from pointcloudsegmentation_amd.params import flatten_numpy
    from pointcloudsegmentation_amd.synth import make_batch
    from src.models.point2net import Point2Net
    torch.manual_seed(E)
    m = Point2Net(6, E)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(-0.3, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
    params = flatten_numpy({k: v.numpy() for k, v in m.state_dict().items()})
    x = make_batch(2, 200, seed=E)
    want = oracle.forward_eval(x, params, embedding_dim=E)
    m = m.cuda().eval()
    with torch.no_grad():
        emb = m(torch.from_numpy(x).cuda())
    assert_close_to_max(emb.cpu().numpy(), want["emb"], TOL, f"emb E={E}")


def test_batch_independence_and_determinism(model):
    """Eval output of a cloud does not depend on its batch mates and
    repeated calls are bit-identical."""
    from pointcloudsegmentation_amd.synth import make_batch
    x = torch.from_numpy(make_batch(4, 1000, seed=3)).cuda()
    with torch.no_grad():
        a = model(x)
        b = model(x)
        c = model(x[1:2].contiguous())
    assert torch.equal(a, b)
    assert torch.equal(a[1:2], c)


@pytest.mark.parametrize("bad", [float("nan"), -float("nan"), float("inf"), -float("inf")])
def test_non_finite_input_row_stays_inside_its_cloud(model, ref_weights, bad):
    """The contract for a non-finite input value -- the one place where the eval path is NOT the reference's arithmetic:
    ReLU is an integer max on the bit pattern, which turns a NaN with its sign bit set into 0, and the channel maxima
    drop NaNs, where torch.relu / torch.max propagate them (oracle/p2n_oracle.c, `y > 0 ? y : 0`, drops them too: the
    record of the reference's behaviour below is the stock-PyTorch CPU module, the reference's own op sequence).
      reference (torch):  the bad point's NaN reaches the cloud's global feature and through the concat EVERY point
                          of that cloud (all of its embeddings are NaN); other clouds are untouched.
      HIP path:           other clouds are bit-identical to a clean run; every OTHER point of the bad cloud stays
                          finite; what the bad point's own row and the cloud's global feature hold is unspecified
                          (NaN or finite), but never anything out of bounds."""
    from pointcloudsegmentation_amd.synth import make_batch
    from src.models.point2net import Point2Net
    x = make_batch(3, 300, seed=5)
    xq = x.copy()
    xq[1, 17, 0] = bad
    with torch.no_grad():
        clean = model(torch.from_numpy(x).cuda())
        got = model(torch.from_numpy(xq).cuda())
        g_clean = model.encoder(torch.from_numpy(x).cuda())[3]
        g_got = model.encoder(torch.from_numpy(xq).cuda())[3]
    assert torch.equal(got[0], clean[0]) and torch.equal(got[2], clean[2])
    assert torch.equal(g_got[0], g_clean[0]) and torch.equal(g_got[2], g_clean[2])
    others = torch.cat([got[1, :17], got[1, 18:]])
    assert bool(torch.isfinite(others).all())
    assert got.shape == clean.shape
    cpu = Point2Net(6, 64)
    cpu.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref_weights.items()}, strict=True)
    with torch.no_grad():
        want = cpu.eval()(torch.from_numpy(xq)).numpy()
    assert np.isnan(want[1]).all()                       # the reference's behaviour, for the record
    np.testing.assert_array_equal(np.isfinite(want[0]), True)
    assert_close_to_max(got[0].cpu().numpy(), want[0], TOL, "clean cloud beside a poisoned one")


def test_weight_update_invalidates_packed_image(model):
    from pointcloudsegmentation_amd.synth import make_batch
    x = torch.from_numpy(make_batch(1, 100, seed=4)).cuda()
    with torch.no_grad():
        a = model(x).clone()
        model.embedding_net[6].bias.add_(1.0)
        b = model(x)
        model.embedding_net[6].bias.sub_(1.0)
    np.testing.assert_allclose((b - a).cpu().numpy(), 1.0, atol=1e-5)


def test_errors(model):
    with pytest.raises(ValueError):
        model(torch.zeros(2, 10, 5, device="cuda"))
    with pytest.raises(TypeError):
        model(torch.zeros(2, 10, 6, device="cuda", dtype=torch.float64))
    with torch.no_grad():
        assert model(torch.zeros(0, 10, 6, device="cuda")).shape == (0, 10, 64)
