"""Scene-level inference on the GPU: the compute of the reference's
`PointCloudInference.predict` (reference inference.py:143-188) with every stage in
libp2n_hip.so -- input normalisation (:166, csrc/p2n_prep.hip), Point2Net forward (:170,
csrc/p2n_fwd.hip), embedding clustering (:174-178, csrc/p2n_cluster.hip) -- and nothing
leaving the device between them.  File IO, PLY writers and visualisation stay with the caller.
"""
import numpy as np
import torch

from . import runtime as _rt
from .clustering import cluster_embeddings
from .preprocess import embed_scene, fused_ok, preprocess_point_cloud, preprocess_stats


@torch.no_grad()
def predict(model, points, colors=None, eps: float = 0.1, min_samples: int = 50, to_host: bool = True) -> dict:
    """Same arguments and result dict as the reference method: instance_labels (N,) int64,
    embeddings (N,D) float32, num_instances, centroid, scale.  `model` is a Point2Net in eval
    mode on a GPU.  to_host=False keeps labels and embeddings as GPU tensors (the reference
    always returns numpy)."""
    if model.training:
        raise RuntimeError("predict: call model.eval() first (reference inference.py:73)")
    dev = next(model.parameters()).device
    if fused_ok(model) and not (isinstance(points, torch.Tensor) and points.device != dev):
        # the normalisation's affine runs in the encoder's layer-0 loader (p2n_forward_eval_raw): no (N,6) feature tensor
        p, c, cs8 = preprocess_stats(points, colors)
        emb = embed_scene(model, p, c, cs8)                    # (N, D)
        h = cs8[:4].cpu().numpy()
        centroid, scale = h[:3].copy(), np.float64(h[3])
    else:
        features, centroid, scale = preprocess_point_cloud(points, colors)
        if features.device != dev:
            features = features.to(dev)
        emb = model(features)[0]                               # (N, D)
    labels = cluster_embeddings(emb, eps=eps, min_samples=min_samples)
    # reference inference.py:183: len(np.unique(labels)) - 1, whether or not a noise label occurs
    num = int(torch.unique(labels).numel()) - 1
    if to_host:
        labels, emb = _rt.download(labels, emb)   # pinned staging, one synchronisation
    return {"instance_labels": labels, "embeddings": emb, "num_instances": num, "centroid": centroid, "scale": scale}


@torch.no_grad()
def predict_many(model, scenes, eps: float = 0.1, min_samples: int = 50, overlap: bool = True):
    """predict() over a sequence of scenes [(points, colors | None), ...] as a three-stage pipeline on one
    GPU: while scene i runs forward -> cluster on the compute stream, scene i+1's arrays are already being
    uploaded from pinned staging buffers and normalised on a second stream, and scene i-1's labels and
    embeddings leave through pinned buffers on a third (reference inference.py:168-176 does upload, forward,
    download and clustering one after the other on one stream).  Yields the same dict per scene as predict().
    overlap=False runs the scenes one by one through predict() (the comparison tools/bench_inference.py
    reports)."""
    if model.training:
        raise RuntimeError("predict_many: call model.eval() first (reference inference.py:73)")
    scenes = list(scenes)
    if not overlap:
        for pts, col in scenes:
            yield predict(model, pts, col, eps=eps, min_samples=min_samples)
        return
    dev = next(model.parameters()).device
    main = torch.cuda.current_stream(dev)
    up, down = _side_streams(dev)

    # Pinned staging is cached per (device, embedding size) and only ever grows: a fresh pinned allocation waits for
    # the busy GPU (hipHostMalloc synchronises with the device) and costs milliseconds per call -- with a handful of
    # scenes per call that prologue, not the pipeline, was the time per scene.  Uploads use two slots sized for the
    # largest scene seen so far; a slot is rewritten only after its previous upload has completed.
    nmax = max(len(p) for p, _ in scenes) if scenes else 0
    E = model.embedding_dim
    up_ring, ring = _pinned_rings(dev, nmax, E)

    fused = fused_ok(model)

    def stage(k):
        pts, col = scenes[k]
        p_h, c_h, last = up_ring[k % 2]
        if last[0] is not None:
            last[0].synchronize()
        n = len(pts)
        p_h[:n].copy_(torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float64)))
        if col is not None:
            c_h[:n].copy_(torch.from_numpy(np.ascontiguousarray(col, dtype=np.float64)))
        with torch.cuda.stream(up):
            p = p_h[:n].to(dev, non_blocking=True)
            c = c_h[:n].to(dev, non_blocking=True) if col is not None else None
            copied = torch.cuda.Event()
            copied.record(up)
            # the normalisation rides on the upload stream too: its float64 row sum is one dependent chain on one
            # CU (numpy's order), which hides completely beside the previous scene's forward and clustering
            if fused:       # statistics here; the affine itself runs in the encoder's layer-0 loader on the compute stream
                job = preprocess_stats(p, c)
            else:
                job = preprocess_point_cloud(p, c, sync=False)
            ev = torch.cuda.Event()
            ev.record(up)
        last[0] = copied
        return job, ev

    # Results leave through a ring of three pinned slots (same cache) and are copied out into ordinary arrays when
    # yielded.
    def finish(job):
        slot, n, ev = job
        labels_h, emb_h, cs_h, num_h, _ = ring[slot]
        ev.synchronize()
        cs = cs_h.numpy()
        # (copied out by torch: its large copies run on all the host threads, numpy's on one -- at 150,000 points that
        # one-thread copy of 38 MB of embeddings, not the GPU, set the time per scene once the clustering had become fast)
        emb_o = torch.empty((n, emb_h.shape[1]), dtype=emb_h.dtype).copy_(emb_h[:n])
        return {"instance_labels": labels_h[:n].numpy().copy(), "embeddings": emb_o.numpy(),
                "num_instances": int(num_h[0]) - 1, "centroid": cs[:3].copy(), "scale": np.float64(cs[3])}

    pending = None
    nxt = stage(0) if scenes else None
    for i in range(len(scenes)):
        job, ev = nxt
        main.wait_event(ev)
        for t in job:
            if t is not None:
                t.record_stream(main)
        if i + 1 < len(scenes):
            nxt = stage(i + 1)                               # upload + normalisation of scene i+1 beside the compute of scene i
        if fused:
            emb, cs = embed_scene(model, *job), job[2][:4]
        else:
            emb, cs = model(job[0])[0], job[1]
        labels = cluster_embeddings(emb, eps=eps, min_samples=min_samples)
        num = _count_labels(labels)
        n = labels.shape[0]
        done = torch.cuda.Event()
        done.record(main)
        labels_h, emb_h, cs_h, num_h, out_last = ring[i % 3]
        with torch.cuda.stream(down):                        # results leave while scene i+1 computes
            down.wait_event(done)
            labels_h[:n].copy_(labels, non_blocking=True)
            emb_h[:n].copy_(emb, non_blocking=True)
            cs_h.copy_(cs, non_blocking=True)
            num_h.copy_(num.reshape(1), non_blocking=True)
            for t in (labels, emb, cs, num):
                t.record_stream(down)
            out_ev = torch.cuda.Event()
            out_ev.record(down)
        out_last[0] = out_ev
        if pending is not None:
            yield finish(pending)
        pending = (i % 3, n, out_ev)
    if pending is not None:
        yield finish(pending)


_streams = {}
_rings = {}


def _pinned_rings(dev: torch.device, nmax: int, E: int):
    """The pinned upload slots (2) and result slots (3) of predict_many for this device and embedding size, sized
    for the largest scene seen so far.  Every slot carries the event of its last transfer: a call that finds one
    pending (an earlier generator abandoned half-way) waits for it before the slot is reused."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), E)
    ent = _rings.get(key)
    if ent is None or ent[0] < nmax:
        cap = max(nmax, 1)
        up = [(torch.empty((cap, 3), dtype=torch.float64, pin_memory=True),
               torch.empty((cap, 3), dtype=torch.float64, pin_memory=True), [None]) for _ in range(2)]
        out = [(torch.empty(cap, dtype=torch.int64, pin_memory=True), torch.empty((cap, E), dtype=torch.float32, pin_memory=True),
                torch.empty(4, dtype=torch.float64, pin_memory=True), torch.empty(1, dtype=torch.int64, pin_memory=True), [None])
               for _ in range(3)]
        ent = (cap, up, out)
        _rings[key] = ent
    for slot in ent[1]:
        if slot[2][0] is not None:
            slot[2][0].synchronize()
            slot[2][0] = None
    for slot in ent[2]:
        if slot[4][0] is not None:
            slot[4][0].synchronize()
            slot[4][0] = None
    return ent[1], ent[2]


def _side_streams(dev: torch.device):
    """The upload and download streams of a device, created once: the caching allocator keeps a pool per
    stream, so fresh streams per call would pay hipMalloc for every staging tensor again."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _streams:
        _streams[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _streams[key]


def _count_labels(labels: torch.Tensor) -> torch.Tensor:
    """number of distinct labels as a device scalar (no host synchronisation): labels are 0 (noise) or
    1..K without gaps (reference inference.py:131-140 renumbers them), so it is max + 1 when a 0 occurs and
    max otherwise -- the quantity the reference reports as len(np.unique(labels))."""
    mx = labels.max().to(torch.int64)
    has0 = (labels == 0).any().to(torch.int64)
    return mx + has0


class PointCloudInference:
    """Same constructor and methods as the reference class (reference inference.py:43-219), with the
    compute on libp2n_hip.so.  The device must be a GPU: the reference silently falls back to the CPU
    (:53); here a missing GPU or library raises, because a silent CPU path is exactly what this
    package is not.  The checkpoint is read with weights_only=True (it holds tensors and plain
    Python containers)."""

    def __init__(self, checkpoint_path: str, config_path: str, device: str = "cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PointCloudInference runs on a GPU (device='cuda'); use the reference path for CPUs")
        self.config = self.load_config(config_path)
        self.model = self.load_model(checkpoint_path)
        from .io import HighPrecisionLabeler
        self.labeler = HighPrecisionLabeler()

    def load_config(self, config_path: str) -> dict:
        import yaml
        with open(config_path, "r") as f:
            return yaml.safe_load(f)

    def load_model(self, checkpoint_path: str) -> torch.nn.Module:
        from .point2net import Point2Net
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        model = Point2Net(feature_dim=6, embedding_dim=self.config["model"]["embedding_dim"]).to(self.device)
        model.load_state_dict(checkpoint["model_state_dict"])
        model.eval()
        return model

    def preprocess_point_cloud(self, points, colors=None):
        """reference :76-104 -> (features (1,N,6) float32 on the GPU, centroid, scale)."""
        return preprocess_point_cloud(points, colors)

    def cluster_embeddings(self, embeddings, eps: float = 0.1, min_samples: int = 50):
        """reference :106-142; accepts a numpy array or a GPU tensor, returns int64 labels as numpy."""
        t = torch.as_tensor(embeddings)
        if t.device.type != "cuda":
            t = t.to(self.device)
        return cluster_embeddings(t, eps=eps, min_samples=min_samples).cpu().numpy()

    def predict(self, points, colors=None, eps: float = 0.1, min_samples: int = 50) -> dict:
        return predict(self.model, points, colors, eps=eps, min_samples=min_samples)

    def save_results(self, points, colors, instance_labels, output_path: str):
        from .io import save_results
        save_results(points, colors, instance_labels, output_path)

    def run(self, input_path: str, output_path: str, eps: float = 0.1, min_samples: int = 50) -> dict:
        """What the reference's main() does between argument parsing and logging (:254-277): read the
        PLY, predict, write the three result files."""
        from .io import read_point_cloud_ply
        points, colors, _ = read_point_cloud_ply(input_path)
        results = self.predict(points, colors, eps=eps, min_samples=min_samples)
        self.save_results(points, colors, results["instance_labels"], output_path)
        return results
