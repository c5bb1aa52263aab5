"""Accel-34: ResNet-101 key branch + pre-activation ResNet-34 correction branch
(reference dff_deeplab/symbols/accel_34.py)."""
from .accel_base import _basic_branch


class accel_34(_basic_branch):
    version = '34'
    branch_prefix = '34_'
    r_units = [3, 4, 6]
    conv5_units = 3
