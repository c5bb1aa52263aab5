"""Shared test-time graph assembly of the four Accel models.

Reference: dff_deeplab/symbols/accel_18.py:121-239, accel_34.py:121-239,
accel_50.py:116-228, accel_101.py:104-193.  Inputs are `data`, `data_key`,
`feat_key`; the key graph outputs [data_key, feat_key, res5c_relu_output,
croped_score_output], the cur graph [data_key, warping_feat_output,
correction_output] (Accel-101: ..., croped_score_output).
"""
from .. import mx
from .resnet_v1_101_flownet_deeplab import resnet_v1_101_flownet_deeplab


class accel_base(resnet_v1_101_flownet_deeplab):
    version = None          # '18' | '34' | '50' | '101'
    branch_prefix = None    # parameter prefix of the R-branch head ('18_', '34_', 'curr_')

    def __init__(self):
        resnet_v1_101_flownet_deeplab.__init__(self)

    # DeepLab task head: fc6 1x1 -> ReLU -> score 1x1 -> 32x32/16 grouped Deconvolution -> Crop(8,8)
    def _task_head(self, feat, data, num_classes, p=''):
        fc6 = mx.symbol.Convolution(
            data=feat, kernel=(1, 1), pad=(0, 0), num_filter=1024, name=p + 'fc6',
            bias=mx.symbol.Variable(p + 'fc6_bias', lr_mult=2.0),
            weight=mx.symbol.Variable(p + 'fc6_weight', lr_mult=1.0), workspace=self.workspace)
        relu_fc6 = mx.sym.Activation(data=fc6, act_type='relu', name=p + 'relu_fc6')
        score = mx.symbol.Convolution(
            data=relu_fc6, kernel=(1, 1), pad=(0, 0), num_filter=num_classes, name=p + 'score',
            bias=mx.symbol.Variable(p + 'score_bias', lr_mult=2.0),
            weight=mx.symbol.Variable(p + 'score_weight', lr_mult=1.0), workspace=self.workspace)
        upsampling = mx.symbol.Deconvolution(
            data=score, num_filter=num_classes, kernel=(32, 32), stride=(16, 16), num_group=num_classes,
            no_bias=True, name=p + 'upsampling', attr={'lr_mult': '0.0'}, workspace=self.workspace)
        return mx.symbol.Crop(*[upsampling, data], offset=(8, 8), name=p + 'croped_score')

    def _correction(self, stacked_in, num_filter):
        # named `correction`, parameters `corr_weight` / `corr_bias` (accel_18.py:232-235)
        return mx.symbol.Convolution(
            data=stacked_in, kernel=(1, 1), pad=(0, 0), num_filter=num_filter, name='correction',
            bias=mx.symbol.Variable('corr_bias', lr_mult=4.0),
            weight=mx.symbol.Variable('corr_weight', lr_mult=2.0), workspace=self.workspace)

    def _r_branch_features(self, data_cur):
        raise NotImplementedError()

    def get_key_test_symbol(self, cfg):
        # cfg.CLASS_AGNOSTIC / cfg.network.NUM_ANCHORS are read-but-unused detection leftovers in
        # the reference (accel_18.py:124-126); ignoring them lets the training YAMLs drop in too.
        num_classes = cfg.dataset.NUM_CLASSES
        data = mx.sym.Variable(name='data')
        data_key = mx.sym.Variable(name='data_key')
        feat_key = mx.sym.Variable(name='feat_key')
        conv_feat = self.get_resnet_dcn(data)
        croped_score = self._task_head(conv_feat, data, num_classes)
        group = mx.sym.Group([data_key, feat_key, conv_feat, croped_score])
        self.sym = group
        return group

    def get_cur_test_symbol(self, cfg):
        num_classes = cfg.dataset.NUM_CLASSES
        data_cur = mx.sym.Variable(name='data')
        data_key = mx.sym.Variable(name='data_key')
        conv_feat = mx.sym.Variable(name='feat_key')

        flow, scale_map = self.get_flownet(data_cur, data_key)   # scale_map is dead here (ref F7)
        flow_grid = mx.sym.GridGenerator(data=flow, transform_type='warp', name='flow_grid')
        conv_feat = mx.sym.BilinearSampler(data=conv_feat, grid=flow_grid, name='warping_feat')

        if self.version == '101':
            # feature-level fusion (accel_101.py:161-191)
            feat_curr = self.get_resnet_dcn(data_cur)
            stacked_in = mx.sym.Concat(*[conv_feat, feat_curr], dim=1)
            feat_fuse = self._correction(stacked_in, 2048)
            croped_score = self._task_head(feat_fuse, data_cur, num_classes)
            group = mx.sym.Group([data_key, conv_feat, croped_score])
        else:
            # score-level fusion (accel_18.py:177-235)
            croped_score = self._task_head(conv_feat, data_cur, num_classes)
            feat_curr = self._r_branch_features(data_cur)
            curr_croped_score = self._task_head(feat_curr, data_cur, num_classes, self.branch_prefix)
            stacked_in = mx.sym.Concat(*[croped_score, curr_croped_score], dim=1)
            correction = self._correction(stacked_in, num_classes)
            group = mx.sym.Group([data_key, conv_feat, correction])
        self.sym = group
        return group

    def get_train_symbol(self, cfg):
        raise NotImplementedError("training graphs are outside the inference hot path (SURVEY.md 8f rank 4)")

    def get_batch_test_symbol(self, cfg):
        raise NotImplementedError("R-FCN detection leftover; needs MultiProposal/PSROIPooling (out of scope)")

    def init_weight(self, cfg, arg_params, aux_params):
        pass


class _basic_branch(accel_base):
    units = None
    conv5_units = None

    def _r_branch_features(self, data_cur):
        p = self.branch_prefix
        feat = self.resnet(data_sym=data_cur, prefix=p, units=self.r_units, num_stages=3,
                           filter_list=[64, 64, 128, 256, 512], num_classes=1000, data_type='imagenet',
                           bottle_neck=False, bn_mom=0.9, workspace=512, memonger=False)
        feat = self._basic_dcn_conv5(feat, p, self.conv5_units)
        return mx.symbol.Deconvolution(data=feat, num_filter=2048, kernel=(4, 4), stride=(2, 2), pad=(1, 1),
                                       no_bias=True, name=p + 'feat_upsampling', workspace=self.workspace,
                                       attr={'lr_mult': '2.0'})
