"""Thin host runtime over the C ABI: streams, workspaces, pointer plumbing.

PyTorch is used for device memory and streams only; every kernel on a GPU tensor
goes through libp2n_hip.so (see _lib.py) and there is no fallback.
"""
from typing import Dict, Optional, Tuple

import torch

from . import _lib

_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}


def stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Per (device, stream) scratch buffer, grown on demand and reused: kernels on one
    stream run in order, so consecutive calls can share it."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_handle(device))
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def upload(a, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """numpy array -> GPU tensor through a pinned staging buffer (torch's caching host allocator
    hands the block back on later calls), so the copy is one DMA instead of the driver's pageable
    path; ordered on the current stream."""
    import numpy as np
    src = torch.from_numpy(np.ascontiguousarray(a))
    stage = torch.empty(src.shape, dtype=dtype, pin_memory=True)
    stage.copy_(src)                       # host-side conversion + copy into pinned memory
    return stage.to(device, non_blocking=True)


def download(*tensors: torch.Tensor):
    """GPU tensors -> numpy arrays backed by pinned host memory: all copies are queued, then one
    synchronisation.  The arrays own their pinned blocks (returned to torch's host cache when they die)."""
    outs = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    if tensors:
        torch.cuda.current_stream(tensors[0].device).synchronize()
    return tuple(h.numpy() for h in outs)


def require_cuda_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


PRECISIONS = ("fp32", "bf16", "fp32x3")
# precision -> (packed size, pack, workspace size, forward) entry points of the C ABI
_EVAL_ENTRY = {
    "fp32": ("p2n_packed_param_count", "p2n_pack_eval_params", "p2n_forward_eval_workspace_bytes", "p2n_forward_eval"),
    "bf16": ("p2n_packed_param_count_bf16", "p2n_pack_eval_params_bf16", "p2n_forward_eval_workspace_bytes",
             "p2n_forward_eval_bf16"),
    "fp32x3": ("p2n_packed_param_count_x3", "p2n_pack_eval_params_x3", "p2n_forward_eval_x3_workspace_bytes",
               "p2n_forward_eval_x3"),
}


def pack_eval_params(raw: torch.Tensor, feature_dim: int, embedding_dim: int, precision: str = "fp32") -> torch.Tensor:
    """raw flat float32 params on the GPU -> packed eval image (see include/p2n.h)."""
    L = _lib.lib()
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
    count, pack = getattr(L, _EVAL_ENTRY[precision][0]), getattr(L, _EVAL_ENTRY[precision][1])
    n = count(feature_dim, embedding_dim)
    if n < 0:
        raise ValueError(f"unsupported feature_dim={feature_dim} / embedding_dim={embedding_dim} "
                         "(HIP path supports 1..8 / 1..256)")
    if raw.numel() != L.p2n_param_count(feature_dim, embedding_dim):
        raise ValueError("raw parameter buffer has the wrong size")
    raw = require_cuda_f32(raw, "raw")
    packed = torch.empty(n, dtype=torch.float32, device=raw.device)
    with torch.cuda.device(raw.device):
        _lib.check(pack(raw.data_ptr(), feature_dim, embedding_dim, packed.data_ptr(), stream_handle(raw.device)),
                   "p2n_pack_eval_params")
    return packed


def forward_eval(x: torch.Tensor, packed: torch.Tensor, embedding_dim: int, want_global: bool = False,
                 want_features: bool = False, want_embedding: bool = True, precision: str = "fp32"):
    """x (B,N,F) -> emb (B,N,E) [, global_f (B,256)] [, f1, f2, f3].  `packed` must have been built
    for the same precision."""
    L = _lib.lib()
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
    ws_bytes, fwd = getattr(L, _EVAL_ENTRY[precision][2]), getattr(L, _EVAL_ENTRY[precision][3])
    x = require_cuda_f32(x, "x")
    if x.dim() != 3:
        raise ValueError(f"x must be (B,N,C), got {tuple(x.shape)}")
    B, N, F = x.shape
    E = embedding_dim
    dev = x.device
    emb = torch.empty((B, N, E), dtype=torch.float32, device=dev) if want_embedding else None
    g = torch.empty((B, 256), dtype=torch.float32, device=dev) if (want_global or want_features) else None
    f1 = torch.empty((B, 64, N), dtype=torch.float32, device=dev) if want_features else None
    f2 = torch.empty((B, 128, N), dtype=torch.float32, device=dev) if want_features else None
    f3 = torch.empty((B, 256, N), dtype=torch.float32, device=dev) if want_features else None
    if B * N > 0:
        nbytes = ws_bytes(B, N, F, E)
        ws = workspace(nbytes, dev)
        with torch.cuda.device(dev):
            _lib.check(fwd(x.data_ptr(), B, N, F, E, packed.data_ptr(), ptr(emb), ptr(g), ptr(f1), ptr(f2), ptr(f3),
                           ws.data_ptr(), ws.numel(), stream_handle(dev)), "p2n_forward_eval")
    return emb, g, f1, f2, f3
