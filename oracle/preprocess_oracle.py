"""CPU restatement of the reference's input normalisation.  TEST INFRASTRUCTURE ONLY.

Follows reference inference.py:76-104 (PointCloudInference.preprocess_point_cloud)
operation for operation in numpy float64; the reference function cannot be imported in
the build container (inference.py needs open3d), so this parity is pinned by construction
(same numpy calls), not by a reference-generated fixture.
"""
import numpy as np


def preprocess_point_cloud(points, colors=None):
    points = np.asarray(points)
    centroid = np.mean(points, axis=0)                    # inference.py:88
    points = points - centroid                            # :89
    scale = np.max(np.abs(points))                        # :90
    points = points / scale                               # :91
    if colors is None:
        colors = np.ones_like(points) * 0.5               # :94-95
    if colors.max() > 1:
        colors = colors / 255.0                           # :98-99
    features = np.concatenate([points, colors], axis=1)   # :102
    return features.astype(np.float32)[None], centroid, scale   # :103 (.float().unsqueeze(0))
