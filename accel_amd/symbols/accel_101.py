"""Accel-101: ResNet-101 on every frame, 4096->2048 feature fusion
(reference dff_deeplab/symbols/accel_101.py:144-193)."""
from .accel_base import accel_base


class accel_101(accel_base):
    version = '101'
    branch_prefix = None
