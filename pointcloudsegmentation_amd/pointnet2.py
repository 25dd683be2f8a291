"""Hierarchical (PointNet++ style) backbone assembled from the MI355X operators.

BASELINE.json's north star names set-abstraction levels (FPS -> ball query -> grouping ->
shared MLP -> max) and three-NN feature propagation; the reference model has none of them
(reference src/models/point2net.py:5-87 is a pointwise MLP + global max, SURVEY.md section
0), so this class is NOT the drop-in `Point2Net` -- that one keeps matching the reference
and its checkpoint -- but an optional sibling with the same call contract:
`Point2NetSA(feature_dim=6, embedding_dim=64)(x:(B,N,6)) -> (B,N,64)`.

GPU tensors in eval mode run libp2n_hip.so end to end (p2n_fps, p2n_ball_query,
p2n_sa_group, p2n_pointwise_layer, p2n_group_max, p2n_three_nn, p2n_three_interpolate).
CPU tensors run a stock-PyTorch composition of the same formulas (same index rules: exact
fp32 squared distances, lowest index first), which is what the tests compare against
(parity unpinned by the reference).  In train() mode GPU tensors take the per-operator composition with
train-mode shared MLPs (p2n_mlp_train_forward / p2n_mlp_train_backward: batch-statistics BatchNorm, weight and data
gradients as fp32 MFMA GEMMs) and the operators' adjoints (p2n_group_gather_bwd, p2n_three_interpolate_bwd); the max
over a group and the concatenations go through autograd.
"""
import ctypes
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from . import ops
from . import runtime as _rt


# ------------------------------------------------------------------ torch (CPU) operators
def _sqdist(a, b):
    d = a.unsqueeze(2) - b.unsqueeze(1)                     # (B,M,N,3)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


# Indices and interpolation weights are always computed from the float32 coordinates, also when a float64 copy of
# the module runs as a low-noise reference for the gradients: the index rules are defined on fp32 distances.
def fps_torch(xyz, npoint, start_idx=0):
    xyz = xyz.float()
    B, N, _ = xyz.shape
    idx = torch.zeros(B, npoint, dtype=torch.long)
    mind = torch.full((B, N), float("inf"))
    cur = torch.full((B,), start_idx, dtype=torch.long)
    ar = torch.arange(B)
    for m in range(npoint):
        idx[:, m] = cur
        c = xyz[ar, cur].unsqueeze(1)
        d = xyz - c
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        mind = torch.minimum(mind, d2)
        cur = torch.argmax(mind, dim=1)
    return idx


def ball_query_torch(xyz, new_xyz, radius, nsample):
    B, N, _ = xyz.shape
    d2 = _sqdist(new_xyz.float(), xyz.float())
    r2 = torch.tensor(radius, dtype=torch.float32) * torch.tensor(radius, dtype=torch.float32)
    cand = torch.where(d2 < r2, torch.arange(N).view(1, 1, N), torch.tensor(N))
    cand = cand.sort(dim=2).values[:, :, :nsample]
    first = cand[:, :, :1]
    first = torch.where(first == N, torch.zeros_like(first), first)
    return torch.where(cand == N, first.expand_as(cand), cand)


def three_nn_torch(unknown, known):
    d2 = _sqdist(unknown.float(), known.float())
    order = torch.argsort(d2, dim=2, stable=True)[:, :, :3]
    d = torch.gather(d2, 2, order)
    r = 1.0 / (d + 1e-8)
    w = r / ((r[..., 0] + r[..., 1]) + r[..., 2]).unsqueeze(-1)
    return order, w


def _gather_points(feat, idx):                              # feat (B,C,N), idx (B,...) -> (B,C,...)
    B, C, _ = feat.shape
    flat = idx.reshape(B, 1, -1).expand(B, C, -1)
    return torch.gather(feat, 2, flat).reshape(B, C, *idx.shape[1:])


# ------------------------------------------------------------------ fused levels (p2n_sa_level / p2n_fp_level)
class _LevelImage:
    """Packed weight image of one fused level's MLP chain (p2n_level_pack), rebuilt only when a
    parameter or BatchNorm buffer of the chain changed."""

    def __init__(self, kind: int):
        self.kind, self.key, self.packed, self.dims = kind, None, None, None

    @staticmethod
    def chain(mlps):
        """[(conv, bn | None), ...] of the SharedMLP modules in `mlps`, in order."""
        out = []
        for mlp in mlps:
            mods = list(mlp.net)
            i = 0
            while i < len(mods):
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
                out.append((mods[i], bn))
                i += 3 if bn is not None else 1
        return out

    def supported(self, mlps, rot: int) -> bool:
        ch = self.chain(mlps)
        dims = [ch[0][0].in_channels] + [c.out_channels for c, _ in ch]
        want_plain_last = self.kind == 2
        for i, (_, bn) in enumerate(ch):
            plain = bn is None
            if plain != (want_plain_last and i == len(ch) - 1):
                return False
            if bn is not None and (bn.eps != ch[0][1].eps or not bn.track_running_stats or not bn.affine):
                return False
        arr = (ctypes.c_int * len(dims))(*dims)
        return _lib.lib().p2n_level_packed_floats(self.kind, len(ch), arr, rot) >= 0

    def get(self, mlps, rot: int, device) -> torch.Tensor:
        ch = self.chain(mlps)
        tensors = []
        for conv, bn in ch:
            tensors += [conv.weight, conv.bias]
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [None] * 4
        # running statistics are also updated through raw pointers (the train-mode kernels), which bumps no version
        # counter: num_batches_tracked, incremented in place next to every such update, stands in for them
        nbt = tuple(bn.num_batches_tracked._version for _, bn in ch if bn is not None and bn.num_batches_tracked is not None)
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors) + nbt
        if key != self.key:
            L = _lib.lib()
            dims = [ch[0][0].in_channels] + [c.out_channels for c, _ in ch]
            arr = (ctypes.c_int * len(dims))(*dims)
            n = L.p2n_level_packed_floats(self.kind, len(ch), arr, rot)
            if n < 0:
                raise ValueError(f"no fused level kernel for the chain {dims}")
            for t in tensors:
                if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise ValueError("level parameters must be contiguous float32 GPU tensors")
            table = (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
            packed = torch.empty(n, dtype=torch.float32, device=device)
            with torch.cuda.device(device):
                _lib.check(L.p2n_level_pack(self.kind, len(ch), arr, table, rot, float(ch[0][1].eps), packed.data_ptr(),
                                            _rt.stream_handle(device)), "p2n_level_pack")
            self.packed, self.key, self.dims = packed, key, dims
        return self.packed


def gather_xyz(rows: torch.Tensor, ld: int, idx: torch.Tensor) -> torch.Tensor:
    """rows (B,N,ld) with the coordinates in the first three columns, idx (B,M) int32 -> (B,M,3)."""
    B, N = rows.shape[0], rows.shape[1]
    M = idx.shape[1]
    out = torch.empty((B, M, 3), dtype=torch.float32, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(_lib.lib().p2n_gather_xyz(rows.data_ptr(), ld, idx.data_ptr(), B, N, M, out.data_ptr(),
                                             _rt.stream_handle(rows.device)), "p2n_gather_xyz")
    return out


# ------------------------------------------------------------------ shared MLP
def _to_cm(x: torch.Tensor, Ppad: int) -> torch.Tensor:
    """(B,C,L) -> channel-major (C, Ppad) over the flattened batch, pad columns zero"""
    B, C, L = x.shape
    out = torch.zeros((C, Ppad), dtype=torch.float32, device=x.device) if Ppad != B * L else \
        torch.empty((C, Ppad), dtype=torch.float32, device=x.device)
    out[:, :B * L].view(C, B, L).copy_(x.permute(1, 0, 2))
    return out


def _from_cm(m: torch.Tensor, B: int, L: int) -> torch.Tensor:
    """channel-major (C, Ppad) -> (B,C,L) contiguous"""
    C = m.shape[0]
    return m[:, :B * L].view(C, B, L).permute(1, 0, 2).contiguous()


class _SharedMLPTrainFn(torch.autograd.Function):
    """Train-mode shared MLP on GPU tensors: p2n_mlp_train_forward / p2n_mlp_train_backward (batch-statistics
    BatchNorm with the running buffers updated in place, as nn.BatchNorm1d in train() mode)."""

    @staticmethod
    def forward(ctx, x, dims, last_plain, momentum, buffers, group, *params):
        L_ = _lib.lib()
        x = _rt.require_cuda_f32(x, "x")
        B, C, L = x.shape
        P = B * L
        Ppad = (P + 127) // 128 * 128
        nl = len(dims) - 1
        dev = x.device
        arr = (ctypes.c_int * len(dims))(*dims)
        nbytes = L_.p2n_mlp_train_workspace_bytes(P, nl, arr, int(last_plain))
        if nbytes == 0:
            raise ValueError("p2n_mlp_train_workspace_bytes: " + L_.p2n_last_error().decode(errors="replace"))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # holds the activations until the backward
        x_cm = _to_cm(x, Ppad)
        out_cm = torch.empty((dims[-1], Ppad), dtype=torch.float32, device=dev)
        table, pi = [], 0
        for l in range(nl):
            plain = last_plain and l == nl - 1
            w, b = params[pi], params[pi + 1]
            pi += 2
            if plain:
                table += [w, b, None, None, None, None]
            else:
                g, be = params[pi], params[pi + 1]
                pi += 2
                table += [w, b, g, be, buffers[2 * l], buffers[2 * l + 1]]
        for t in table:
            if t is not None and not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("shared-MLP parameters and buffers must be contiguous float32 tensors on the input's GPU")
        ptrs = (ctypes.c_void_p * len(table))(*[None if t is None else t.data_ptr() for t in table])
        with torch.cuda.device(dev):
            _lib.check(L_.p2n_mlp_train_forward(x_cm.data_ptr(), P, nl, arr, int(last_plain), ptrs, float(momentum),
                                                out_cm.data_ptr(), ws.data_ptr(), ws.numel(), _rt.stream_handle(dev)),
                       "p2n_mlp_train_forward")
        ctx.meta = (dims, bool(last_plain), B, L, P, Ppad, table, group)
        ctx.done = False
        if group:
            # maximum over groups of `group` consecutive columns (a set-abstraction level's neighbours) taken on the
            # channel-major output: the (B,C,M,ns) tensor, torch.max and its scatter backward never exist
            if L % group:
                raise ValueError(f"columns per cloud ({L}) are not a multiple of the group size ({group})")
            G = P // group
            mx = torch.empty((dims[-1], G), dtype=torch.float32, device=dev)
            arg = torch.empty((dims[-1], G), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L_.p2n_group_max_cm(out_cm.data_ptr(), dims[-1], Ppad, G, group, mx.data_ptr(), arg.data_ptr(),
                                               _rt.stream_handle(dev)), "p2n_group_max_cm")
            ctx.save_for_backward(x_cm, ws, arg, *params)
            return _from_cm(mx, B, L // group)
        ctx.save_for_backward(x_cm, ws, None, *params)
        return _from_cm(out_cm, B, L)

    @staticmethod
    def backward(ctx, gout):
        if ctx.done:
            raise RuntimeError("the shared MLP's saved activations were consumed by a previous backward "
                               "(retain_graph is not supported on this path)")
        ctx.done = True
        L_ = _lib.lib()
        dims, last_plain, B, L, P, Ppad, table, group = ctx.meta
        x_cm, ws, arg = ctx.saved_tensors[0], ctx.saved_tensors[1], ctx.saved_tensors[2]
        params = ctx.saved_tensors[3:]
        nl = len(dims) - 1
        dev = x_cm.device
        arr = (ctypes.c_int * len(dims))(*dims)
        if group:
            G = P // group
            g_cm = _to_cm(_rt.require_cuda_f32(gout, "grad_output"), G)        # (B,C,M) -> (C, B*M)
            dout_cm = torch.empty((dims[-1], Ppad), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L_.p2n_group_max_cm_bwd(g_cm.data_ptr(), arg.data_ptr(), dims[-1], G, group, Ppad, dout_cm.data_ptr(),
                                                   _rt.stream_handle(dev)), "p2n_group_max_cm_bwd")
        else:
            dout_cm = _to_cm(_rt.require_cuda_f32(gout, "grad_output"), Ppad)  # a private copy: the call overwrites it
        grads = [torch.empty_like(p) for p in params]
        gtab, gi = [], 0
        for l in range(nl):
            plain = last_plain and l == nl - 1
            if plain:
                gtab += [grads[gi], grads[gi + 1], None, None]
                gi += 2
            else:
                gtab += grads[gi:gi + 4]
                gi += 4
        need_dx = ctx.needs_input_grad[0]
        dx_cm = torch.empty((dims[0], Ppad), dtype=torch.float32, device=dev) if need_dx else None
        ptrs = (ctypes.c_void_p * len(table))(*[None if t is None else t.data_ptr() for t in table])
        gptrs = (ctypes.c_void_p * len(gtab))(*[None if t is None else t.data_ptr() for t in gtab])
        with torch.cuda.device(dev):
            _lib.check(L_.p2n_mlp_train_backward(x_cm.data_ptr(), dout_cm.data_ptr(), P, nl, arr, int(last_plain), ptrs,
                                                 gptrs, _rt.ptr(dx_cm), ws.data_ptr(), ws.numel(), _rt.stream_handle(dev)),
                       "p2n_mlp_train_backward")
        dx = _from_cm(dx_cm, B, L) if need_dx else None
        return (dx, None, None, None, None, None, *grads)


def shared_mlp_train(net: nn.Sequential, x: torch.Tensor, group: int = 0, eval_stats: bool = False) -> torch.Tensor:
    """Train-mode forward, with autograd, of nn.Sequential(Conv1d(k=1), BatchNorm1d, ReLU, ... [, Conv1d(k=1)]) on a
    GPU tensor (B,C,L) through p2n_mlp_train_forward / p2n_mlp_train_backward; BatchNorm buffers are updated as
    nn.BatchNorm1d does in train() mode.  group > 0: followed by the maximum over groups of `group` consecutive
    columns -> (B, Cout, L // group), i.e. net(x).view(B, Cout, L // group, group).max(dim=3).values.
    eval_stats: eval-mode BatchNorm under autograd -- running statistics used, nothing updated."""
    mods, i = list(net), 0
    if not mods or not isinstance(mods[0], nn.Conv1d):
        raise NotImplementedError("a shared MLP starts with a Conv1d")
    dims, params, buffers, bns, last_plain = [mods[0].in_channels], [], [], [], False
    while i < len(mods):
        conv = mods[i]
        if not isinstance(conv, nn.Conv1d) or conv.kernel_size != (1,) or conv.bias is None:
            raise NotImplementedError("the GPU shared MLP is built from Conv1d(kernel_size=1, bias=True) layers")
        dims.append(conv.out_channels)
        params += [conv.weight, conv.bias]
        if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):
            bn = mods[i + 1]
            if not (i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)):
                raise NotImplementedError("every BatchNorm1d of a GPU shared MLP is followed by a ReLU")
            if bn.momentum is None and not eval_stats:
                raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative average) is not implemented on the GPU path")
            if abs(bn.eps - 1e-5) > 1e-12 or not bn.affine or not bn.track_running_stats:
                raise NotImplementedError("the GPU shared MLP needs BatchNorm1d(eps=1e-5, affine=True, track_running_stats=True)")
            params += [bn.weight, bn.bias]
            buffers += [bn.running_mean, bn.running_var]
            bns.append(bn)
            i += 3
        else:
            if i + 1 != len(mods):
                raise NotImplementedError("only the last layer of a shared MLP may be a plain Conv1d")
            last_plain = True
            buffers += [None, None]
            i += 1
    if eval_stats:
        return _SharedMLPTrainFn.apply(x, tuple(dims), last_plain, -1.0, tuple(buffers), int(group), *params)
    if len({bn.momentum for bn in bns}) > 1:
        raise NotImplementedError("the BatchNorm layers of one shared MLP must share one momentum")
    momentum = bns[0].momentum if bns else 0.1
    y = _SharedMLPTrainFn.apply(x, tuple(dims), last_plain, momentum, tuple(buffers), int(group), *params)
    with torch.no_grad():
        for bn in bns:
            bn.num_batches_tracked += 1
    return y


class SharedMLP(nn.Module):
    """[Conv1d(k=1) + BatchNorm1d + ReLU] x n over (B,C,L).  GPU tensors: eval = p2n_pointwise_layer (or the
    fused level kernels), train / autograd = p2n_mlp_train_forward + p2n_mlp_train_backward."""

    def __init__(self, channels: List[int], last_plain: bool = False):
        super().__init__()
        layers = []
        for i in range(len(channels) - 1):
            plain = last_plain and i == len(channels) - 2
            layers.append(nn.Conv1d(channels[i], channels[i + 1], 1))
            if not plain:
                layers += [nn.BatchNorm1d(channels[i + 1]), nn.ReLU()]
        self.net = nn.Sequential(*layers)

    def _folded(self):
        out, mods, i = [], list(self.net), 0
        while i < len(mods):
            conv = mods[i]
            w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).to(torch.float32).contiguous()
            b = conv.bias.detach().double()
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):
                bn = mods[i + 1]
                s = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
                t = (b - bn.running_mean.double()) * s + bn.bias.detach().double()
                out.append((w, s.float().contiguous(), t.float().contiguous(), 1))
                i += 3
            else:
                out.append((w, torch.ones_like(b).float(), b.float().contiguous(), 0))
                i += 1
        return out

    def _forward_train(self, x, group: int = 0):
        """with autograd: batch statistics in train() mode, the running statistics (left untouched) in eval() mode"""
        return shared_mlp_train(self.net, x, group, eval_stats=not self.training)

    def needs_autograd(self, x) -> bool:
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, x):                                   # (B,C,L)
        if not x.is_cuda:
            return self.net(x)
        if self.training or self.needs_autograd(x):
            return self._forward_train(x)
        L = _lib.lib()
        x = _rt.require_cuda_f32(x, "x")
        B, _, n = x.shape
        for w, s, t, relu in self._folded():
            cout, cin = w.shape
            y = torch.empty((B, cout, n), dtype=torch.float32, device=x.device)
            ws = _rt.workspace(L.p2n_pointwise_layer_workspace_bytes(cin, cout), x.device)
            with torch.cuda.device(x.device):
                _lib.check(L.p2n_pointwise_layer(x.data_ptr(), B, cin, n, w.data_ptr(), s.data_ptr(), t.data_ptr(), relu,
                                                 cout, y.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _rt.stream_handle(x.device)), "p2n_pointwise_layer")
            x = y
        return x


class SetAbstraction(nn.Module):
    """FPS -> ball query -> grouping (relative xyz ++ features) -> shared MLP -> max over the group."""

    def __init__(self, npoint_div: int, radius: float, nsample: int, in_channels: int, mlp: List[int]):
        super().__init__()
        self.npoint_div, self.radius, self.nsample = npoint_div, radius, nsample
        self.mlp = SharedMLP([in_channels + 3] + mlp)
        self._image = _LevelImage(0)

    def fused_supported(self) -> bool:
        return self.nsample == 32 and len(self.mlp.net) == 9 and self._image.supported([self.mlp], 3)

    def forward_rows(self, xyz, rows, ld, feat_off, C, geometry=None):
        """The level on point-major rows, one fused kernel (p2n_sa_level): xyz (B,N,3) contiguous;
        rows (B,N,ld) whose first three columns are the coordinates when feat_off > 0 (level 1 reads the
        input tensor itself: x = [xyz | features]) -- otherwise rows are the C features alone and the
        coordinates come from xyz.  Returns new_xyz (B,M,3), features (B,M,Cout) point-major."""
        L = _lib.lib()
        B, N, _ = xyz.shape
        M = max(1, N // self.npoint_div)
        dev = xyz.device
        if geometry is None:
            fidx = ops.farthest_point_sample(xyz, M)
            new_xyz = gather_xyz(xyz, 3, fidx)
            gidx = ops.ball_query(xyz, new_xyz, self.radius, self.nsample)
        else:                                           # (new_xyz, gidx) computed ahead on the geometry streams
            new_xyz, gidx = geometry
        packed = self._image.get([self.mlp], 3, dev)
        mlp = self._image.dims[1:]
        out = torch.empty((B, M, mlp[-1]), dtype=torch.float32, device=dev)
        if feat_off > 0:
            xyz_ptr, ldx, feat_ptr, ldf = rows.data_ptr(), ld, rows.data_ptr() + 4 * feat_off, ld
        else:
            xyz_ptr, ldx, feat_ptr, ldf = xyz.data_ptr(), 3, (rows.data_ptr() if C else None), ld
        with torch.cuda.device(dev):
            _lib.check(L.p2n_sa_level(xyz_ptr, ldx, new_xyz.data_ptr(), feat_ptr if C else None, ldf, gidx.data_ptr(),
                                      B, C, N, M, self.nsample, (ctypes.c_int * 3)(*mlp), packed.data_ptr(),
                                      out.data_ptr(), _rt.stream_handle(dev)), "p2n_sa_level")
        return new_xyz, out

    def forward(self, xyz, feat: Optional[torch.Tensor]):   # xyz (B,N,3), feat (B,C,N) | None
        B, N, _ = xyz.shape
        M = max(1, N // self.npoint_div)
        ns = self.nsample
        if xyz.is_cuda:
            L = _lib.lib()
            fidx = ops.farthest_point_sample(xyz, M)
            xyz_t = xyz.transpose(1, 2).contiguous()
            new_xyz = ops.group_gather(xyz_t, fidx.unsqueeze(2)).squeeze(3).transpose(1, 2).contiguous()
            gidx = ops.ball_query(xyz, new_xyz, self.radius, ns)
            C = 0 if feat is None else feat.size(1)
            if torch.is_grad_enabled() and (self.training or self.mlp.needs_autograd(xyz if feat is None else feat)):
                # differentiable composition: relative coordinates (no gradient) ++ gathered features (p2n_group_gather
                # with its scatter-add adjoint), train-mode shared MLP, max over the group through autograd
                rel = torch.empty((B, 3, M, ns), dtype=torch.float32, device=xyz.device)
                with torch.cuda.device(xyz.device):
                    _lib.check(L.p2n_sa_group(xyz.data_ptr(), new_xyz.data_ptr(), None, gidx.data_ptr(), B, 0, N, M, ns,
                                              rel.data_ptr(), _rt.stream_handle(xyz.device)), "p2n_sa_group")
                grouped = rel if feat is None else torch.cat([rel, ops.group_gather(feat.contiguous(), gidx)], dim=1)
                return new_xyz, self.mlp._forward_train(grouped.view(B, C + 3, M * ns), group=ns)   # MLP + max over the group
            grouped = torch.empty((B, C + 3, M, ns), dtype=torch.float32, device=xyz.device)
            with torch.cuda.device(xyz.device):
                _lib.check(L.p2n_sa_group(xyz.data_ptr(), new_xyz.data_ptr(), _rt.ptr(feat), gidx.data_ptr(), B, C, N, M,
                                          ns, grouped.data_ptr(), _rt.stream_handle(xyz.device)), "p2n_sa_group")
            y = self.mlp(grouped.view(B, C + 3, M * ns))
            out = torch.empty((B, y.size(1), M), dtype=torch.float32, device=xyz.device)
            with torch.cuda.device(xyz.device):
                _lib.check(L.p2n_group_max(y.data_ptr(), B, y.size(1), M, ns, out.data_ptr(),
                                           _rt.stream_handle(xyz.device)), "p2n_group_max")
            return new_xyz, out
        fidx = fps_torch(xyz, M)
        new_xyz = torch.gather(xyz, 1, fidx.unsqueeze(2).expand(B, M, 3))
        gidx = ball_query_torch(xyz, new_xyz, self.radius, ns)
        rel = _gather_points(xyz.transpose(1, 2), gidx) - new_xyz.transpose(1, 2).unsqueeze(3)
        grouped = rel if feat is None else torch.cat([rel, _gather_points(feat, gidx)], dim=1)
        y = self.mlp(grouped.reshape(B, grouped.size(1), M * ns)).reshape(B, -1, M, ns)
        return new_xyz, y.max(dim=3).values


class FeaturePropagation(nn.Module):
    """three-NN inverse-distance interpolation of the coarse features, skip concat, shared MLP."""

    def __init__(self, in_channels: int, mlp: List[int]):
        super().__init__()
        self.mlp = SharedMLP([in_channels] + mlp)
        self._image = _LevelImage(1)
        self._image_head = _LevelImage(2)

    def fused_supported(self, C1: int, head: Optional["SharedMLP"] = None) -> bool:
        if head is None:
            return len(self.mlp.net) == 6 and self._image.supported([self.mlp], C1)
        return len(self.mlp.net) == 6 and len(head.net) == 4 and self._image_head.supported([self.mlp, head], C1)

    def forward_rows(self, xyz1, xyz2, skip, ld_skip, skip_off, C1, feat2, head: Optional["SharedMLP"] = None,
                     neighbours=None):
        """The level on point-major rows, one fused kernel (p2n_fp_level): xyz1 (B,n,3) fine points, xyz2
        (B,m,3) coarse points, skip rows (B,n,ld_skip) whose columns skip_off..skip_off+C1-1 are the fine
        points' own features (None with C1 = 0), feat2 (B,m,C2) coarse features.  With `head` the embedding
        head runs on top inside the same kernel.  Returns (B,n,Cout) point-major."""
        L = _lib.lib()
        B, n, _ = xyz1.shape
        m, C2 = feat2.shape[1], feat2.shape[2]
        dev = xyz1.device
        idx3, w3 = neighbours if neighbours is not None else ops.three_nn(xyz1, xyz2)[1:]
        image, mlps = (self._image, [self.mlp]) if head is None else (self._image_head, [self.mlp, head])
        packed = image.get(mlps, C1, dev)
        mlp = image.dims[1:]
        out = torch.empty((B, n, mlp[-1]), dtype=torch.float32, device=dev)
        skip_ptr = skip.data_ptr() + 4 * skip_off if C1 else None
        mlp_arr = (ctypes.c_int * len(mlp))(*mlp)
        nbytes = L.p2n_fp_level_workspace_bytes(B, n, C1, C2, len(mlp), mlp_arr)
        ws = _rt.workspace(nbytes, dev) if nbytes else None
        with torch.cuda.device(dev):
            _lib.check(L.p2n_fp_level(idx3.data_ptr(), w3.data_ptr(), feat2.data_ptr(), skip_ptr, ld_skip, B, n, m, C1,
                                      C2, len(mlp), mlp_arr, packed.data_ptr(), out.data_ptr(), _rt.ptr(ws),
                                      ws.numel() if ws is not None else 0, _rt.stream_handle(dev)), "p2n_fp_level")
        return out

    def forward(self, xyz1, xyz2, feat1: Optional[torch.Tensor], feat2):
        if xyz1.is_cuda:
            _, idx, w = ops.three_nn(xyz1, xyz2)
            interp = ops.three_interpolate(feat2, idx, w)
        else:
            idx, w = three_nn_torch(xyz1, xyz2)
            f = _gather_points(feat2, idx)                  # (B,C,n,3)
            interp = (w[:, None, :, 0] * f[..., 0] + w[:, None, :, 1] * f[..., 1]) + w[:, None, :, 2] * f[..., 2]
        x = interp if feat1 is None else torch.cat([feat1, interp], dim=1)
        return self.mlp(x.contiguous())


class Point2NetSA(nn.Module):
    """Three set-abstraction levels, three feature-propagation levels, embedding head."""

    def __init__(self, feature_dim: int = 6, embedding_dim: int = 64, nsample: int = 32,
                 radii=(0.1, 0.2, 0.4)):
        super().__init__()
        c = feature_dim - 3
        self.feature_dim, self.embedding_dim = feature_dim, embedding_dim
        self.fused = True        # GPU eval: fused level kernels (p2n_sa_level / p2n_fp_level); False = per-operator composition
        self.overlap_geometry = True   # fused path: FPS chain / ball queries / three-NN on side streams beside the level kernels
        self.sa1 = SetAbstraction(4, radii[0], nsample, c, [32, 32, 64])
        self.sa2 = SetAbstraction(4, radii[1], nsample, 64, [64, 64, 128])
        self.sa3 = SetAbstraction(4, radii[2], nsample, 128, [128, 128, 256])
        self.fp3 = FeaturePropagation(256 + 128, [256, 256])
        self.fp2 = FeaturePropagation(256 + 64, [256, 128])
        self.fp1 = FeaturePropagation(128 + c, [128, 128])
        self.head = SharedMLP([128, 128, embedding_dim], last_plain=True)

    def forward(self, x):                                   # (B,N,feature_dim) -> (B,N,E)
        if x.dim() != 3 or x.size(2) != self.feature_dim:
            raise ValueError(f"expected x of shape (B, N, {self.feature_dim}), got {tuple(x.shape)}")
        if x.is_cuda and self.fused and self._fused_ok():
            return self._forward_fused(x)
        xyz0 = x[:, :, :3].contiguous()
        f0 = x[:, :, 3:].transpose(1, 2).contiguous() if self.feature_dim > 3 else None
        xyz1, f1 = self.sa1(xyz0, f0)
        xyz2, f2 = self.sa2(xyz1, f1)
        xyz3, f3 = self.sa3(xyz2, f2)
        u2 = self.fp3(xyz2, xyz3, f2, f3)
        u1 = self.fp2(xyz1, xyz2, f1, u2)
        u0 = self.fp1(xyz0, xyz1, f0, u1)
        if x.is_cuda and ops.fps_needs_check(x.shape[0], x.shape[1]):
            ops.fps_check(x.device)
        return self.head(u0).transpose(1, 2)

    # ---- GPU eval: one fused kernel per level, features point-major between levels -------------------
    def _fused_ok(self) -> bool:
        if self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return False      # training / autograd: the per-operator path (train-mode shared MLPs, differentiable operators)
        c = self.feature_dim - 3
        return (self.sa1.fused_supported() and self.sa2.fused_supported() and self.sa3.fused_supported()
                and self.fp3.fused_supported(self.sa2.mlp.net[-3].out_channels)
                and self.fp2.fused_supported(self.sa1.mlp.net[-3].out_channels)
                and self.fp1.fused_supported(c, self.head))

    def _forward_fused(self, x):
        """Everything that depends on coordinates only -- the FPS chain (65 % of the forward, latency-bound on
        16 of the 256 CUs), the ball queries and the three-NN searches -- runs on side streams beside the level
        kernels: FPS of level l+1 starts as soon as level l's centroids exist, not after level l's MLP."""
        x = _rt.require_cuda_f32(x, "x")
        F = self.feature_dim
        c = F - 3
        dev = x.device
        main = torch.cuda.current_stream(dev)
        xyz0 = x[:, :, :3].contiguous()
        # Clouds sampled by clusters of fat WORKGROUPS (> 160,000 points) are planned as if they owned every CU: beside the
        # level kernels of other streams their workgroups may not all be resident and the exchange would time out.  They
        # run the whole forward in order on the caller's stream instead (no overlap, nothing to time out against).
        # Wave-member clusters (4,096..160,000 points) are a few registers and an LDS sliver per member: they fit beside
        # the level kernels and keep the overlap; either kind is checked where the indices have been consumed.
        clusters = ops.fps_needs_check(x.shape[0], x.shape[1])
        if not self.overlap_geometry or ops.fps_needs_whole_chip(x.shape[0], x.shape[1]):
            xyz1, f1 = self.sa1.forward_rows(xyz0, x, F, 3, c)                     # level 1 reads x's rows in place
            xyz2, f2 = self.sa2.forward_rows(xyz1, f1, f1.shape[2], 0, f1.shape[2])
            xyz3, f3 = self.sa3.forward_rows(xyz2, f2, f2.shape[2], 0, f2.shape[2])
            u2 = self.fp3.forward_rows(xyz2, xyz3, f2, f2.shape[2], 0, f2.shape[2], f3)
            u1 = self.fp2.forward_rows(xyz1, xyz2, f1, f1.shape[2], 0, f1.shape[2], u2)
            emb = self.fp1.forward_rows(xyz0, xyz1, x, F, 3, c, u1, head=self.head)   # (B,N,E)
        else:
            start = torch.cuda.Event()
            start.record(main)
            emb = self._levels(x, xyz0, self._geometry(xyz0, start))
        if clusters:
            ops.fps_check(x.device)      # a timed-out exchange (another process's kernels on the GPU) raises here
        return emb


    def _geometry(self, xyz0, start, lane: int = 0):
        """Everything that depends on coordinates only, queued on the three geometry streams behind `start`:
        centroids of the three levels (FPS chain), ball-query groups, three-NN neighbours and weights.  Returns the
        tensors and the events the level kernels wait for."""
        dev = xyz0.device
        main = torch.cuda.current_stream(dev)
        s_fps, s_search, s_nn = _geometry_streams(dev, lane)
        sas = (self.sa1, self.sa2, self.sa3)
        for st in (s_fps, s_search, s_nn):
            xyz0.record_stream(st)
        xyz, cent = [xyz0], []
        with torch.cuda.stream(s_fps):                   # fps1 -> fps2 -> fps3, nothing in between
            s_fps.wait_event(start)
            for sa in sas:
                M = max(1, xyz[-1].shape[1] // sa.npoint_div)
                nx = gather_xyz(xyz[-1], 3, ops.farthest_point_sample(xyz[-1], M))
                ev = torch.cuda.Event()
                ev.record(s_fps)
                xyz.append(nx)
                cent.append(ev)
        with torch.cuda.stream(s_nn):                    # the largest search (fine points x level-1 centroids)
            s_nn.wait_event(cent[0])
            nn01 = ops.three_nn(xyz[0], xyz[1])[1:]
            e_nn01 = torch.cuda.Event()
            e_nn01.record(s_nn)
        groups, e_ball = [], []
        with torch.cuda.stream(s_search):
            for l, sa in enumerate(sas):
                s_search.wait_event(cent[l])
                groups.append(ops.ball_query(xyz[l], xyz[l + 1], sa.radius, sa.nsample))
                ev = torch.cuda.Event()
                ev.record(s_search)
                e_ball.append(ev)
            nn23 = ops.three_nn(xyz[2], xyz[3])[1:]
            nn12 = ops.three_nn(xyz[1], xyz[2])[1:]
            e_nn = torch.cuda.Event()
            e_nn.record(s_search)
        for t in xyz[1:] + groups + list(nn01) + list(nn12) + list(nn23):
            t.record_stream(main)                        # allocated on a side stream, consumed on the caller's
        xyz[1].record_stream(s_nn)
        for t in xyz[1:]:
            t.record_stream(s_search)
        return {"xyz": xyz, "groups": groups, "nn01": nn01, "nn12": nn12, "nn23": nn23,
                "e_ball": e_ball, "e_nn": e_nn, "e_nn01": e_nn01}

    def _levels(self, x, xyz0, geo):
        """The six fused level kernels on the caller's stream, each behind the event of the geometry it needs."""
        F = self.feature_dim
        c = F - 3
        main = torch.cuda.current_stream(x.device)
        xyz, groups = geo["xyz"], geo["groups"]
        main.wait_event(geo["e_ball"][0])
        _, f1 = self.sa1.forward_rows(xyz0, x, F, 3, c, geometry=(xyz[1], groups[0]))
        main.wait_event(geo["e_ball"][1])
        _, f2 = self.sa2.forward_rows(xyz[1], f1, f1.shape[2], 0, f1.shape[2], geometry=(xyz[2], groups[1]))
        main.wait_event(geo["e_ball"][2])
        _, f3 = self.sa3.forward_rows(xyz[2], f2, f2.shape[2], 0, f2.shape[2], geometry=(xyz[3], groups[2]))
        main.wait_event(geo["e_nn"])
        u2 = self.fp3.forward_rows(xyz[2], xyz[3], f2, f2.shape[2], 0, f2.shape[2], f3, neighbours=geo["nn23"])
        u1 = self.fp2.forward_rows(xyz[1], xyz[2], f1, f1.shape[2], 0, f1.shape[2], u2, neighbours=geo["nn12"])
        main.wait_event(geo["e_nn01"])
        return self.fp1.forward_rows(xyz0, xyz[1], x, F, 3, c, u1, head=self.head, neighbours=geo["nn01"])

    def forward_many(self, batches):
        """Embeddings of a sequence of batches [(B,N,F), ...] already on the GPU, as a pipeline over them: the
        sampling chains of the next batches (latency-bound on a few CUs, 65 % of a lone forward) run beside the level
        kernels of batch i.  GPU eval only (the fused path); results equal forward(batch) bit for bit."""
        batches = list(batches)
        if not batches:
            return []
        x0 = batches[0]
        if not (x0.is_cuda and self.fused and self._fused_ok()):
            return [self(x) for x in batches]
        for x in batches:
            if x.dim() != 3 or x.size(2) != self.feature_dim:
                raise ValueError(f"expected batches of shape (B, N, {self.feature_dim}), got {tuple(x.shape)}")
        if any(ops.fps_needs_whole_chip(x.shape[0], x.shape[1]) for x in batches):
            return [self._forward_fused(x) for x in batches]   # workgroup clusters must own the chip: no pipelining
        dev = x0.device
        main = torch.cuda.current_stream(dev)
        xs = [_rt.require_cuda_f32(x, "x") for x in batches]
        xyz0 = [x[:, :, :3].contiguous() for x in xs]
        start = torch.cuda.Event()                       # every batch is resident: its geometry waits for nothing else
        start.record(main)
        # GEO_LANES sampling chains run side by side (one uses B of the 256 CUs), each on its own stream triple
        out, geo = [], {}
        for i in range(min(GEO_LANES, len(xs))):
            geo[i] = self._geometry(xyz0[i], start, i % GEO_LANES)
        for i, x in enumerate(xs):
            out.append(self._levels(x, xyz0[i], geo.pop(i)))
            j = i + GEO_LANES
            if j < len(xs):
                geo[j] = self._geometry(xyz0[j], start, j % GEO_LANES)
        if any(ops.fps_needs_check(x.shape[0], x.shape[1]) for x in xs):
            ops.fps_check(dev)       # wave clusters sampled these clouds: one check (and synchronisation) for the whole sequence
        return out


_geo_streams = {}
GEO_LANES = 3      # geometry pipelines in flight in forward_many


def _geometry_streams(dev: torch.device, lane: int = 0):
    """three side streams per device and lane (FPS chain, ball queries + small three-NN searches, the large
    three-NN search), created once: the caching allocator keeps a pool per stream"""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), lane)
    if key not in _geo_streams:
        _geo_streams[key] = tuple(torch.cuda.Stream(device=dev) for _ in range(3))
    return _geo_streams[key]
