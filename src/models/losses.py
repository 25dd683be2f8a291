"""Drop-in for the reference's `src/models/losses.py` import path (reference train.py:19)."""
from pointcloudsegmentation_amd.losses import (DiscriminativeLoss, FocalLoss, InstanceSegLoss,  # noqa: F401
                                               SmoothnessLoss)
