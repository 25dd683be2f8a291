"""Host-side mirror of the reference model module (reference src/models/point2net.py).

Same class names, constructor arguments, sub-module names and therefore the same
58 state_dict keys, so the reference checkpoint loads with
strict=True and `train.py` / `inference.py`-shaped callers work unchanged.

Where the arithmetic runs:
  * GPU tensors: always libp2n_hip.so (hand-written gfx950 kernels).  A missing
    library raises; nothing silently falls back.
  * CPU tensors: the sub-modules ARE stock torch.nn layers, so the module runs
    as ordinary PyTorch on the host -- the behaviour reference inference.py:53
    relies on when no GPU is present.  This is also bench.py's "torch-cpu" leg.
"""
import torch
import torch.nn as nn

from . import params as _params
from . import runtime as _rt


# What a freshly constructed module computes in on GPU tensors.  "fp32": exact-fp32 MFMA (the reference's dtype).
# "fp32x3": fp32-accurate products from exact three-way bf16 splits on the bf16 matrix pipe (same parity bar).
# "bf16": bf16 operands, 2e-2.  tests/test_*_x3_gpu.py flip this to run the fp32 suites unchanged in fp32x3 mode.
DEFAULT_PRECISION = "fp32"


def _mlp(cin, cmid, cout):
    # reference point2net.py:9-16 (and :18-34): Conv1d(k=1) -> BatchNorm1d -> ReLU, twice
    return nn.Sequential(
        nn.Conv1d(cin, cmid, 1), nn.BatchNorm1d(cmid), nn.ReLU(),
        nn.Conv1d(cmid, cout, 1), nn.BatchNorm1d(cout), nn.ReLU())


class _PackedCache:
    """Eval-mode packed weight image, rebuilt only when a parameter/buffer changed."""

    def __init__(self):
        self.key = None
        self.packed = None
        self._perm = None
        self._count = -1

    def get(self, module: nn.Module, feature_dim: int, embedding_dim: int, prefix: str = "",
            precision: str = "fp32") -> torch.Tensor:
        """prefix="" : module is a Point2Net.  prefix="encoder.": module is a bare
        PointNet2Encoder; the head slots of the image are zero-filled (never read:
        the head kernel is skipped when no embedding is requested)."""
        # hot path: this runs on every forward.  parameters()/buffers() walk the module tree in a fixed
        # order; the permutation into flat (layer-major) order is worked out once per module structure.
        ps = list(module.parameters()) + list(module.buffers())
        if self._perm is None or self._count != len(ps):
            names = [n for n, _ in module.named_parameters()] + [n for n, _ in module.named_buffers()]
            order = _params.flat_order()
            if prefix:
                order = [k[len(prefix):] for k in order if k.startswith(prefix)]
            self._perm = [names.index(k) for k in order]
            self._count = len(ps)
        tensors = [ps[i] for i in self._perm]
        # the train-mode kernels update running_mean / running_var through raw pointers (no version bump); every such
        # update increments num_batches_tracked in place, so its version stands in -- also when a sub-module (the bare
        # encoder) was trained on its own and this cache belongs to its parent
        nbt = tuple(b._version for b in ps if b.dtype == torch.int64)
        key = (precision,) + tuple([(t.data_ptr(), t._version) for t in tensors]) + nbt
        if key != self.key:
            parts = [t.detach().reshape(-1).to(torch.float32) for t in tensors]
            have = sum(p.numel() for p in parts)
            need = _params.param_count(feature_dim, embedding_dim)
            if have < need:
                parts.append(torch.zeros(need - have, dtype=torch.float32, device=parts[0].device))
            self.packed = _rt.pack_eval_params(torch.cat(parts), feature_dim, embedding_dim, precision)
            self.key = key
        return self.packed


class PointNet2Encoder(nn.Module):
    """reference point2net.py:5-48."""

    def __init__(self, in_channels=6):
        super().__init__()
        self.mlp1 = _mlp(in_channels, 64, 64)
        self.mlp2 = _mlp(64, 128, 128)
        self.mlp3 = _mlp(128, 256, 256)
        self.in_channels = in_channels
        self.precision = DEFAULT_PRECISION   # "fp32" | "fp32x3" | "bf16" (GPU tensors only)
        self._cache = _PackedCache()

    def forward(self, x):
        if x.is_cuda:
            if self.training or (torch.is_grad_enabled() and
                                 (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
                from .train import encode_train   # noqa: WPS433  (kept out of the inference import path)
                return encode_train(self, x)
            packed = self._cache.get(self, self.in_channels, 32, prefix="encoder.", precision=self.precision)
            _, g, f1, f2, f3 = _rt.forward_eval(x, packed, 32, want_features=True, want_embedding=False,
                                                precision=self.precision)
            return f1, f2, f3, g
        # CPU: stock PyTorch, same op sequence as reference point2net.py:36-48
        x = x.transpose(1, 2)
        f1 = self.mlp1(x)
        f2 = self.mlp2(f1)
        f3 = self.mlp3(f2)
        global_f = torch.max(f3, dim=2)[0]
        return f1, f2, f3, global_f


class Point2Net(nn.Module):
    """reference point2net.py:50-87: forward(x:(B,N,C)) -> (B,N,embedding_dim)."""

    def __init__(self, feature_dim=6, embedding_dim=64):
        super().__init__()
        self.feature_dim = feature_dim
        self.embedding_dim = embedding_dim
        self.encoder = PointNet2Encoder(feature_dim)
        self.embedding_net = nn.Sequential(
            nn.Conv1d(256 + 256, 256, 1), nn.BatchNorm1d(256), nn.ReLU(),
            nn.Conv1d(256, 128, 1), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, embedding_dim, 1))
        # Not in the reference: "fp32" (default, exact-fp32 MFMA, 1e-5 parity), "fp32x3" (three-way bf16 split
        # products with fp32 accumulation on the bf16 MFMA: fp32 accuracy, same 1e-5 parity, eval and training) or
        # "bf16" (bf16 operands, 2e-2 parity).  Plain attribute: not part of the state_dict.
        self.precision = DEFAULT_PRECISION
        self._cache = _PackedCache()

    # ---- GPU paths -----------------------------------------------------
    def _needs_grad(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _gpu_eval(self, x, want_global=False, want_features=False):
        packed = self._cache.get(self, self.feature_dim, self.embedding_dim, precision=self.precision)
        return _rt.forward_eval(x, packed, self.embedding_dim, want_global, want_features, precision=self.precision)

    def forward(self, x):
        if x.dim() != 3 or x.size(2) != self.feature_dim:
            raise ValueError(f"expected x of shape (B, N, {self.feature_dim}), got {tuple(x.shape)}")
        if x.is_cuda and x.dtype != torch.float32:
            raise TypeError(f"the HIP path computes in float32, got {x.dtype}")
        if not x.is_cuda:
            # CPU: stock PyTorch ops, reference point2net.py:75-87
            f1, f2, f3, global_f = self.encoder(x)
            global_f = global_f.unsqueeze(2).expand(-1, -1, f3.size(2))
            combined = torch.cat([f3, global_f], dim=1)
            return self.embedding_net(combined).transpose(1, 2)
        if self.training or self._needs_grad(x):
            from .train import forward_train   # noqa: WPS433
            return forward_train(self, x)
        return self._gpu_eval(x)[0]
