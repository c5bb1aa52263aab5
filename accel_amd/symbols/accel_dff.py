"""Deep Feature Flow only (README "Main Results" row "DFF": R101 key frames +
FlowNet propagation, no correction branch).  The reference tree ships no
symbol file for this row; this is the Accel cur graph with the R branch and the
fusion conv removed, i.e. accel_18.py:161-197 followed directly by the output
group -- provided so the same harness reproduces that row."""
from .. import mx
from .accel_base import accel_base


class accel_dff(accel_base):
    version = 'dff'
    branch_prefix = None

    def get_cur_test_symbol(self, cfg):
        num_classes = cfg.dataset.NUM_CLASSES
        data_cur = mx.sym.Variable(name='data')
        data_key = mx.sym.Variable(name='data_key')
        conv_feat = mx.sym.Variable(name='feat_key')
        flow, _ = self.get_flownet(data_cur, data_key)
        flow_grid = mx.sym.GridGenerator(data=flow, transform_type='warp', name='flow_grid')
        conv_feat = mx.sym.BilinearSampler(data=conv_feat, grid=flow_grid, name='warping_feat')
        croped_score = self._task_head(conv_feat, data_cur, num_classes)
        group = mx.sym.Group([data_key, conv_feat, croped_score])
        self.sym = group
        return group
