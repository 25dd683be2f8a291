"""Batch producer for the training / inference loops.

The reference's `src.data.scannet_dataset.ScanNetDataset` is imported by train.py:15 but is NOT in
the reference repository, so only its contract is known -- **parity unpinned**:
  * constructor `ScanNetDataset(files=..., config=..., num_points=...)` (reference train.py:393-403);
  * scene files hold `points` float32 (N,3) normalised to the unit cube, `colors` float32 (N,3),
    `instance_labels` int32 (N,) (reference src/utils/data_utils.py:79-82, HDF5);
  * a collated batch is a dict with `features` (B,num_points,6) float32, `instance_labels`
    (B,num_points) and `scene_id` (reference train.py:118-120,186).
This module provides exactly that, plus the split helper of reference train.py:335-376 and a
device-side producer that resamples every scene to `num_points` with the FPS operator and uploads
batch i+1 through pinned memory on a side stream while batch i is being computed.

Scene files: `.npz` with the three arrays above always work; `.h5` files (the reference's format)
need h5py, which this image does not have -- opening one without it raises ImportError.
"""
from pathlib import Path
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


def create_train_val_split(data_dir, train_ratio: float = 0.8, seed: int = 42, pattern: str = "*.h5"):
    """reference train.py:335-376: sorted file list, RandomState(seed).shuffle, split at
    int(len * train_ratio).  `pattern` defaults to the reference's '*.h5'; '*.npz' for this
    module's own scene files."""
    data_dir = Path(data_dir)
    files = sorted(data_dir.glob(pattern))
    if not files:
        raise RuntimeError(f"No preprocessed files ({pattern}) found in {data_dir}! Please run preprocessing first.")
    arr = np.array(files)
    np.random.RandomState(seed).shuffle(arr)
    k = int(len(arr) * train_ratio)
    return arr[:k].tolist(), arr[k:].tolist()


def load_scene(path) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> points (N,3) float32, colors (N,3) float32, instance_labels (N,) int32."""
    path = Path(path)
    if path.suffix == ".npz":
        with np.load(path) as z:
            pts, col, lab = z["points"], z["colors"], z["instance_labels"]
    elif path.suffix in (".h5", ".hdf5"):
        try:
            import h5py
        except ImportError as e:      # not installed in this image
            raise ImportError(f"{path}: reading the reference's HDF5 scenes needs h5py") from e
        with h5py.File(path, "r") as f:
            pts, col, lab = f["points"][()], f["colors"][()], f["instance_labels"][()]
    else:
        raise ValueError(f"{path}: unknown scene file type")
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    col = np.ascontiguousarray(col, dtype=np.float32).reshape(-1, 3)
    lab = np.ascontiguousarray(lab, dtype=np.int32).reshape(-1)
    if not (len(pts) == len(col) == len(lab)):
        raise ValueError(f"{path}: {len(pts)} points, {len(col)} colours, {len(lab)} labels")
    return pts, col, lab


def save_scene_npz(path, points, colors, instance_labels) -> None:
    """The reference's scene schema (data_utils.py:79-82) as an .npz file."""
    np.savez(path, points=np.asarray(points, np.float32), colors=np.asarray(colors, np.float32),
             instance_labels=np.asarray(instance_labels, np.int32))


def _scene_features(pts, col, use_color: bool, normalize_color: bool) -> np.ndarray:
    if not use_color:
        col = np.zeros_like(pts)
    elif normalize_color and col.size and col.max() > 1.0:
        col = col / 255.0
    return np.concatenate([pts, col.astype(np.float32)], axis=1)


class ScanNetDataset(Dataset):
    """Map-style dataset with the constructor the reference calls (train.py:393-403).  Every item is
    one scene resampled to `num_points` by a seeded random choice (without replacement when the
    scene has enough points); use `DeviceBatchProducer` for FPS resampling on the GPU."""

    def __init__(self, files: Sequence, config: Optional[dict] = None, num_points: int = 8192, seed: int = 0):
        self.files = [Path(f) for f in files]
        cfg = (config or {}).get("dataset", {}) if isinstance(config, dict) else {}
        self.use_color = bool(cfg.get("use_color", True))
        self.normalize_color = bool(cfg.get("normalize_color", True))
        self.num_points = int(num_points)
        self.seed = int(seed)
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return len(self.files)

    def __getitem__(self, i: int) -> dict:
        pts, col, lab = load_scene(self.files[i])
        n = len(pts)
        if n == 0:
            raise ValueError(f"{self.files[i]}: empty scene")
        rng = np.random.default_rng([self.seed, self.epoch, i])
        sel = rng.choice(n, self.num_points, replace=n < self.num_points)
        feats = _scene_features(pts[sel], col[sel], self.use_color, self.normalize_color)
        return {"features": torch.from_numpy(feats), "instance_labels": torch.from_numpy(lab[sel].astype(np.int64)),
                "scene_id": self.files[i].stem}


class DeviceBatchProducer:
    """Yields the same batch dicts with `features` / `instance_labels` already on `device`.

    Per scene: the whole scene is uploaded, `num_points` of its points are chosen by the FPS operator
    (csrc/p2n_ops.hip; start index drawn from the seeded generator; scenes with fewer points than
    `num_points` are padded by a seeded random choice with replacement) and gathered on the device.
    Uploads run on a side stream from pinned staging buffers, one batch ahead of the consumer: while
    the caller computes on batch i, batch i+1 is being copied and sampled.  `resample='random'`
    replaces FPS by the dataset's random choice (host side)."""

    def __init__(self, files: Sequence, config: Optional[dict] = None, num_points: int = 8192, batch_size: int = 8,
                 device="cuda", shuffle: bool = False, seed: int = 0, resample: str = "fps", drop_last: bool = False):
        if resample not in ("fps", "random"):
            raise ValueError("resample must be 'fps' or 'random'")
        self.ds = ScanNetDataset(files, config, num_points, seed)
        self.batch_size, self.shuffle, self.seed, self.resample, self.drop_last = batch_size, shuffle, seed, resample, drop_last
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatchProducer needs a GPU device")
        self.epoch = 0
        self._side = None       # one side stream per producer (its kernels' workspaces are keyed by stream)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)
        self.ds.set_epoch(epoch)

    def __len__(self) -> int:
        n = len(self.ds)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self) -> List[int]:
        idx = np.arange(len(self.ds))
        if self.shuffle:
            np.random.default_rng([self.seed, self.epoch]).shuffle(idx)
        return idx.tolist()

    def _stage(self, scene_ids: List[int], stream: torch.cuda.Stream) -> dict:
        """Upload and resample one batch on `stream`; returns the batch dict plus the event that
        marks it ready and the pinned buffers that must outlive the copies."""
        from . import ops
        P = self.ds.num_points
        feats, labs, names, keep = [], [], [], []
        check = False
        with torch.cuda.stream(stream):
            for i in scene_ids:
                if self.resample == "random":
                    item = self.ds[i]
                    f_h, l_h = item["features"].pin_memory(), item["instance_labels"].pin_memory()
                    keep += [f_h, l_h]
                    feats.append(f_h.to(self.device, non_blocking=True))
                    labs.append(l_h.to(self.device, non_blocking=True))
                    names.append(item["scene_id"])
                    continue
                pts, col, lab = load_scene(self.ds.files[i])
                n = len(pts)
                if n == 0:
                    raise ValueError(f"{self.ds.files[i]}: empty scene")
                f_h = torch.from_numpy(_scene_features(pts, col, self.ds.use_color, self.ds.normalize_color)).pin_memory()
                l_h = torch.from_numpy(lab.astype(np.int64)).pin_memory()
                keep += [f_h, l_h]
                f_d = f_h.to(self.device, non_blocking=True)
                l_d = l_h.to(self.device, non_blocking=True)
                rng = np.random.default_rng([self.seed, self.epoch, i])
                start = int(rng.integers(0, n))
                m = min(n, P)
                check = check or ops.fps_needs_check(1, n)
                sel = ops.farthest_point_sample(f_d[None, :, :3].contiguous(), m, start)[0].long()
                if m < P:
                    extra = torch.from_numpy(rng.integers(0, n, P - m)).to(self.device)
                    sel = torch.cat([sel, extra])
                feats.append(f_d[sel])
                labs.append(l_d[sel])
                names.append(self.ds.files[i].stem)
            batch = {"features": torch.stack(feats), "instance_labels": torch.stack(labs), "scene_id": names}
            ready = torch.cuda.Event()
            ready.record(stream)
        return {"batch": batch, "ready": ready, "keep": keep, "check": check}

    def __iter__(self) -> Iterator[dict]:
        order = self._order()
        chunks = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        if not chunks:
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        nxt = self._stage(chunks[0], side)
        for k in range(len(chunks)):
            cur = nxt
            # the consumer's stream waits for the staged batch; its tensors were allocated on the side
            # stream, so tell the allocator that the consumer's stream uses them too
            torch.cuda.current_stream(self.device).wait_event(cur["ready"])
            for t in (cur["batch"]["features"], cur["batch"]["instance_labels"]):
                t.record_stream(torch.cuda.current_stream(self.device))
            if cur["check"]:
                # large scenes are resampled by clusters of workgroups that must be co-resident; beside the
                # caller's kernels they may not be: a timed-out exchange is raised here, before the batch is used
                cur["ready"].synchronize()
                from . import ops
                ops.fps_check(self.device)
            if k + 1 < len(chunks):
                nxt = self._stage(chunks[k + 1], side)     # runs ahead while the caller computes on `cur`
            yield cur["batch"]
            cur["ready"].synchronize()                     # pinned staging buffers may go now
