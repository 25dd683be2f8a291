"""Scene-level inference on the GPU: the compute of the reference's
`PointCloudInference.predict` (reference inference.py:143-188) with every stage in
libp2n_hip.so -- input normalisation (:166, csrc/p2n_prep.hip), Point2Net forward (:170,
csrc/p2n_fwd.hip), embedding clustering (:174-178, csrc/p2n_cluster.hip) -- and nothing
leaving the device between them.  File IO, PLY writers and visualisation stay with the caller.
"""
import numpy as np
import torch

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
        labels, emb = labels.cpu().numpy(), emb.cpu().numpy()
    return {"instance_labels": labels, "embeddings": emb, "num_instances": num, "centroid": centroid, "scale": scale}
