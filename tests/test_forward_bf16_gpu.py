"""bf16 eval path (csrc/p2n_fwd_bf16.hip, opt-in via model.precision = "bf16") against the
reference's fp32 outputs: BASELINE north_star allows 2e-2 relative in bf16 (judged, like the
fp32 tolerance, against the tensor's magnitude).  The measured deviation is printed; it should
sit around 6e-3 thanks to the split-bf16 first layers."""
import numpy as np
import pytest
import torch

from conftest import assert_close_to_max, golden

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _model(sd, embedding_dim=64, feature_dim=6):
    from pointcloudsegmentation_amd.point2net import Point2Net
    m = Point2Net(feature_dim=feature_dim, embedding_dim=embedding_dim)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.precision = "bf16"
    m.encoder.precision = "bf16"
    return m


@pytest.fixture(scope="module")
def model(ref_weights):
    return _model(ref_weights)


@pytest.mark.parametrize("name", ["eval_b1_n1_s0", "eval_b1_n7_s1", "eval_b2_n1000_s0", "eval_b3_n4099_s1"])
def test_golden_embeddings_bf16(model, name):
    g = golden(name + ".npz")
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        emb = model(x)
        f1, f2, f3, gf = model.encoder(x)
    err = np.abs(emb.cpu().numpy() - g["emb"]).max() / np.abs(g["emb"]).max()
    print(f"{name}: bf16 deviation {err:.2e}")
    assert_close_to_max(emb.cpu().numpy(), g["emb"], TOL, "emb (bf16)")
    assert_close_to_max(gf.cpu().numpy(), g["global_f"], TOL, "global_f (bf16)")
    if "f3" in g.files:
        assert_close_to_max(f1.cpu().numpy(), g["f1"], TOL, "f1 (bf16)")
        assert_close_to_max(f2.cpu().numpy(), g["f2"], TOL, "f2 (bf16)")
        assert_close_to_max(f3.cpu().numpy(), g["f3"], TOL, "f3 (bf16)")


def test_headline_shape_and_margin(model):
    """BASELINE configs[1] shape: well inside the tolerance (the split first layers buy the margin)."""
    g = golden("eval_b16_n8192_s0.npz")
    from pointcloudsegmentation_amd.synth import make_batch
    x = torch.from_numpy(make_batch(16, 8192, 0)).cuda()
    with torch.no_grad():
        emb = model(x)
    model.precision = "fp32"
    with torch.no_grad():
        ref = model(x)
    model.precision = "bf16"
    err = float((emb - ref).abs().max() / ref.abs().max())
    print(f"bf16 vs fp32 path at 16x8192: {err:.2e}")
    assert err < 1e-2
    assert torch.isfinite(emb).all()


def test_other_embedding_dims_and_ragged():
    from oracle import oracle
    from pointcloudsegmentation_amd import params as P
    from pointcloudsegmentation_amd.point2net import Point2Net
    from pointcloudsegmentation_amd.synth import make_batch
    for E, B, N in [(3, 2, 129), (100, 1, 333), (256, 1, 64), (64, 3, 31)]:
        torch.manual_seed(E)
        m = Point2Net(embedding_dim=E)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
        m = m.cuda().eval()
        x = make_batch(B, N, E)
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        want = oracle.forward_eval(x, P.flatten_numpy(sd), embedding_dim=E)["emb"]
        m.precision = "bf16"
        with torch.no_grad():
            got = m(torch.from_numpy(x).cuda()).cpu().numpy()
        assert_close_to_max(got, want, TOL, f"emb E={E}")


def test_precision_switch_rebuilds_image_and_rejects_unknown(model):
    x = torch.randn(1, 64, 6, device="cuda")
    with torch.no_grad():
        a = model(x)
        model.precision = "fp32"
        b = model(x)
        model.precision = "bf16"
        c = model(x)
    assert torch.equal(a, c) and not torch.equal(a, b)
    model.precision = "fp16"
    with pytest.raises(ValueError):
        with torch.no_grad():
            model(x)
    model.precision = "bf16"
