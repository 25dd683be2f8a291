"""The training path in "fp32x3" mode (p2n_train_forward_x3 / p2n_train_backward_x3: every GEMM operand split exactly
into three bf16 parts, six part products on the bf16 MFMA, fp32 accumulation) against EVERY golden and CPU comparison
the exact-fp32 mode is held to, at the SAME tolerances: the test functions below ARE the fp32 suite's
(tests/test_train_gpu.py, tests/test_loss_gpu.py), collected a second time while freshly constructed modules default
to precision "fp32x3".  Reference goldens: train_*.npz (train.py:118-136 under autograd), gap-safe step vs float64,
the 32 x 8192 noise-floor bar."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _fp32x3_by_default():
    from pointcloudsegmentation_amd import point2net
    old = point2net.DEFAULT_PRECISION
    point2net.DEFAULT_PRECISION = "fp32x3"
    yield
    point2net.DEFAULT_PRECISION = old


def test_modules_really_run_fp32x3():
    from src.models.point2net import Point2Net
    m = Point2Net(6, 64)
    assert m.precision == "fp32x3" and m.encoder.precision == "fp32x3"


from test_train_gpu import (  # noqa: E402,F401
    test_train_forward_backward_sum_loss,
    test_train_small_random_weights_weighted_loss,
    test_train_matches_torch_autograd_on_cpu,
    test_eval_mode_autograd_matches_torch_on_cpu,
    test_train_step_at_full_size_matches_cpu_module,
    test_eval_after_train_forward_without_optimizer_step,
)
from test_loss_gpu import (  # noqa: E402,F401
    test_full_training_step_matches_reference,
    test_full_training_step_gap_safe_cloud,
    test_loss_matches_reference_golden_at_8192,
)
