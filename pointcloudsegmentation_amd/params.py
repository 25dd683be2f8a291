"""Flat parameter layout shared by the C-ABI (include/p2n.h) and the host module.

Layer table of the reference network (reference src/models/point2net.py:9-34
encoder, :57-65 head).  For layer l the flat "raw params" buffer holds
W (Cout,Cin) row-major, bias (Cout), and for the eight BN layers
bn.weight, bn.bias, running_mean, running_var (Cout each).
"""
from typing import Dict, List, Tuple

import numpy as np

# (state_dict prefix of the Conv1d, prefix of its BatchNorm1d or None)
LAYER_KEYS: List[Tuple[str, str]] = [
    ("encoder.mlp1.0", "encoder.mlp1.1"),
    ("encoder.mlp1.3", "encoder.mlp1.4"),
    ("encoder.mlp2.0", "encoder.mlp2.1"),
    ("encoder.mlp2.3", "encoder.mlp2.4"),
    ("encoder.mlp3.0", "encoder.mlp3.1"),
    ("encoder.mlp3.3", "encoder.mlp3.4"),
    ("embedding_net.0", "embedding_net.1"),
    ("embedding_net.3", "embedding_net.4"),
    ("embedding_net.6", None),
]
NLAYER = 9


def layer_dims(feature_dim: int, embedding_dim: int):
    cin = [feature_dim, 64, 64, 128, 128, 256, 512, 256, 128]
    cout = [64, 64, 128, 128, 256, 256, 256, 128, embedding_dim]
    return cin, cout


def param_count(feature_dim: int, embedding_dim: int) -> int:
    cin, cout = layer_dims(feature_dim, embedding_dim)
    return sum(ci * co + co + (4 * co if l < 8 else 0) for l, (ci, co) in enumerate(zip(cin, cout)))


def flat_order() -> List[str]:
    """state_dict keys in flat-buffer order."""
    keys = []
    for conv, bn in LAYER_KEYS:
        keys += [conv + ".weight", conv + ".bias"]
        if bn is not None:
            keys += [bn + ".weight", bn + ".bias", bn + ".running_mean", bn + ".running_var"]
    return keys


def flatten_numpy(sd: Dict[str, np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(sd[k], dtype=np.float32).reshape(-1) for k in flat_order()])
