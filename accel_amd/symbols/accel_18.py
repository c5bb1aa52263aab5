"""Accel-18: ResNet-101 key branch + pre-activation ResNet-18 correction branch
(reference dff_deeplab/symbols/accel_18.py)."""
from .accel_base import _basic_branch


class accel_18(_basic_branch):
    version = '18'
    branch_prefix = '18_'
    r_units = [2, 2, 2]
    conv5_units = 2
