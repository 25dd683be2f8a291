"""Autograd bridge of the training path (libp2n_hip.so: p2n_train_forward/backward).

`Point2Net` in train() mode on GPU tensors -- or in eval() mode with autograd enabled (fine-tuning with
frozen BatchNorm statistics) -- lands here (reference train.py:109,124,131:
model.train(); embeddings = model(features); loss.backward()).  One
torch.autograd.Function covers the whole network: the C library keeps the
pre-BatchNorm activations in a workspace tensor owned by the autograd context and
writes the 34 parameter gradients straight into fresh tensors.
"""
import ctypes

import torch

from . import _lib
from . import params as _params
from . import runtime as _rt


def _param_tensors(model):
    sd = dict(model.named_parameters())
    sd.update(dict(model.named_buffers()))
    return [sd[k] for k in _params.flat_order()]


def _ptr_table(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class _Point2NetTrainFn(torch.autograd.Function):
    """forward(x, model, *learnables) -> emb; grads flow to the 34 learnable tensors."""

    @staticmethod
    def forward(ctx, x, model, momentum, update_stats, *learnables):
        L = _lib.lib()
        B, N, F = x.shape
        E = model.embedding_dim
        dev = x.device
        tensors = _param_tensors(model)
        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("model parameters must be contiguous float32 GPU tensors")
        if not update_stats:
            # running statistics must not move: point the library at scratch copies
            tensors = [t.clone() if ("running" in k) else t for k, t in zip(_params.flat_order(), tensors)]
        nbytes = L.p2n_train_workspace_bytes(B, N, F, E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        emb = torch.empty((B, N, E), dtype=torch.float32, device=dev)
        table = _ptr_table(tensors)
        # model.precision = "bf16": GEMM operands rounded to bf16, fp32 accumulation and storage (opt-in, 2e-2)
        # "fp32x3": exact three-way bf16 splits, six part products, fp32 accumulation: fp32 accuracy on the bf16 matrix pipe
        ctx.suffix = {"fp32": "", "bf16": "_bf16", "fp32x3": "_x3"}[getattr(model, "precision", "fp32")]
        fwd = getattr(L, "p2n_train_forward" + ctx.suffix)
        with torch.cuda.device(dev):
            _lib.check(fwd(x.data_ptr(), B, N, F, E, table, float(momentum), emb.data_ptr(), None,
                           None, None, None, ws.data_ptr(), ws.numel(), _rt.stream_handle(dev)),
                       "p2n_train_forward")
        ctx.model = model
        ctx.ws = ws
        ctx.x = x
        ctx.tensors = tensors     # keeps the (possibly cloned) stats alive for the backward
        ctx.shape = (B, N, F, E)
        # the backward reads the weights through raw pointers: saving them lets autograd's version check
        # catch an in-place update between forward and backward, as it would for stock modules
        ctx.save_for_backward(*learnables)
        return emb

    @staticmethod
    def backward(ctx, gemb):
        L = _lib.lib()
        if ctx.ws is None:
            raise RuntimeError("Trying to backward through the Point2Net HIP graph a second time: the saved "
                               "activations were released after the first backward (retain_graph is not supported)")
        _ = ctx.saved_tensors     # raises if a parameter was modified in place since the forward
        B, N, F, E = ctx.shape
        dev = gemb.device
        gemb = gemb.contiguous()
        order = _params.flat_order()
        learn_keys = [k for k in order if "running" not in k]
        sd = dict(zip(order, ctx.tensors))
        # one flat buffer, the 34 gradients are views of it (each on a 256-byte boundary): autograd installs the
        # views as .grad, and parallel.allreduce_gradients reduces the buffer in place -- no gather, no copy-back
        sizes = [sd[k].numel() for k in learn_keys]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = [flat[o:o + n].view_as(sd[k]) for o, n, k in zip(offs, sizes, learn_keys)]
        bwd = getattr(L, "p2n_train_backward" + ctx.suffix)
        with torch.cuda.device(dev):
            _lib.check(bwd(ctx.x.data_ptr(), gemb.data_ptr(), B, N, F, E, _ptr_table(ctx.tensors),
                           _ptr_table(grads), ctx.ws.data_ptr(), ctx.ws.numel(),
                           _rt.stream_handle(dev)), "p2n_train_backward")
        ctx.ws = None
        return (None, None, None, None) + tuple(grads)


def forward_train(model, x):
    """Train-mode forward of a Point2Net on a GPU tensor (B,N,F) -> (B,N,E) with autograd."""
    eval_stats = not model.training   # eval-mode BatchNorm under autograd: running statistics, left untouched
    if x.requires_grad:
        return forward_train_with_input_grad(model, x)
    x = _rt.require_cuda_f32(x, "x")
    if not eval_stats and x.size(0) * x.size(1) < 2:
        raise ValueError("Expected more than 1 value per channel when training")   # as torch BatchNorm raises
    order = _params.flat_order()
    sd = dict(model.named_parameters())
    learnables = [sd[k] for k in order if "running" not in k]
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    if any(m.momentum is None for m in bns) and not eval_stats:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative moving average) is not implemented on the "
                                  "HIP training path")
    momentum = bns[0].momentum
    if eval_stats:
        return _Point2NetTrainFn.apply(x, model, -1.0, True, *learnables)   # momentum < 0: the library's eval-statistics mode
    emb = _Point2NetTrainFn.apply(x, model, momentum, True, *learnables)
    with torch.no_grad():
        torch._foreach_add_([m.num_batches_tracked for m in bns], 1)      # one launch instead of eight
    # the kernels updated running_mean/var through raw pointers (no version bump): the eval-mode packed
    # images built from the old statistics must not be reused
    model._cache.key = None
    model.encoder._cache.key = None
    return emb


def encode_train(encoder, x):
    """PointNet2Encoder alone under autograd (train() or eval() mode) on a GPU tensor (B,N,F) -> (f1, f2, f3, global_f)
    (reference point2net.py:36-48): the three blocks run as differentiable shared MLPs (p2n_mlp_train_forward /
    p2n_mlp_train_backward, which also return the gradient w.r.t. the input points; batch statistics in train() mode,
    the running statistics in eval() mode), the global max through autograd."""
    from .pointnet2 import shared_mlp_train   # noqa: WPS433
    ev = not encoder.training           # eval() under autograd: running statistics, left untouched
    x = _rt.require_cuda_f32(x, "x")
    if not ev and x.size(0) * x.size(1) < 2:
        raise ValueError("Expected more than 1 value per channel when training")
    xt = x.transpose(1, 2)
    f1 = shared_mlp_train(encoder.mlp1, xt, eval_stats=ev)
    f2 = shared_mlp_train(encoder.mlp2, f1, eval_stats=ev)
    f3 = shared_mlp_train(encoder.mlp3, f2, eval_stats=ev)
    if not ev:
        encoder._cache.key = None       # running statistics changed behind the packed eval image's back
    return f1, f2, f3, torch.max(f3, dim=2)[0]


def forward_train_with_input_grad(model, x):
    """Point2Net (train() or eval() mode) when the input points need a gradient: the reference's op sequence
    (point2net.py:75-87) on train-mode shared MLPs, which produce d/dx; the fused p2n_train_forward path does not."""
    from .pointnet2 import shared_mlp_train   # noqa: WPS433
    f1, f2, f3, g = encode_train(model.encoder, x)
    combined = torch.cat([f3, g.unsqueeze(2).expand(-1, -1, f3.size(2))], dim=1)
    emb = shared_mlp_train(model.embedding_net, combined, eval_stats=not model.training).transpose(1, 2)
    if model.training:
        model._cache.key = None
    return emb
