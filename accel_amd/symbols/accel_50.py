"""Accel-50: ResNet-101 key branch + ResNet-50-DCN correction branch with
`curr_*` head parameters (reference dff_deeplab/symbols/accel_50.py:156-228)."""
from .accel_base import accel_base


class accel_50(accel_base):
    version = '50'
    branch_prefix = 'curr_'

    def _r_branch_features(self, data_cur):
        return self.get_resnet_dcn_50(data_cur)
