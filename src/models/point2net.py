"""Drop-in for the reference's `src/models/point2net.py` import path
(reference train.py:16, inference.py:12): same names, MI355X-native arithmetic."""
from pointcloudsegmentation_amd.point2net import Point2Net, PointNet2Encoder  # noqa: F401
