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
from .preprocess import preprocess_point_cloud


@torch.no_grad()
def predict(model, points, colors=None, eps: float = 0.1, min_samples: int = 50, to_host: bool = True) -> dict:
    """Same arguments and result dict as the reference method: instance_labels (N,) int64,
    embeddings (N,D) float32, num_instances, centroid, scale.  `model` is a Point2Net in eval
    mode on a GPU.  to_host=False keeps labels and embeddings as GPU tensors (the reference
    always returns numpy)."""
    if model.training:
        raise RuntimeError("predict: call model.eval() first (reference inference.py:73)")
    features, centroid, scale = preprocess_point_cloud(points, colors)
    dev = next(model.parameters()).device
    if features.device != dev:
        features = features.to(dev)
    emb = model(features)[0]                                   # (N, D)
    labels = cluster_embeddings(emb, eps=eps, min_samples=min_samples)
    # reference inference.py:183: len(np.unique(labels)) - 1, whether or not a noise label occurs
    num = int(torch.unique(labels).numel()) - 1
    if to_host:
        labels, emb = _rt.download(labels, emb)   # pinned staging, one synchronisation
    return {"instance_labels": labels, "embeddings": emb, "num_instances": num, "centroid": centroid, "scale": scale}


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
