"""Shared graph assembly of the four Accel models.

Reference: dff_deeplab/symbols/accel_18.py:121-239, accel_34.py:121-239,
accel_50.py:116-228, accel_101.py:104-193.  Inputs are `data`, `data_key`,
`feat_key`; the key graph outputs [data_key, feat_key, res5c_relu_output,
croped_score_output], the cur graph [data_key, warping_feat_output,
correction_output] (Accel-101: ..., croped_score_output).

Training graphs (forward only -- the backward pass is out of scope, SURVEY.md 8f
rank 4): accel_18.py:31-119, accel_34.py:31-119, accel_50.py:31-114,
accel_101.py:31-102.  Inputs `data` (1x3xHxW, the labelled frame), `data_ref`
((KEY_INTERVAL-1)x3xHxW, the key frame followed by the intermediate frames),
`eq_flag`, `label`; outputs [softmax_output, data_ref, eq_flag].  The key
frame's ResNet-101 feature is warped KEY_INTERVAL-1 times along FlowNet flows
of consecutive frame pairs (all pairs in ONE FlowNet batch), then corrected by
the branch on `data` exactly as in the test graphs.
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
        """accel_18.py:31-119 (34 / 50 alike) and accel_101.py:31-102."""
        num_classes = cfg.dataset.NUM_CLASSES
        num_interms = cfg.TRAIN.KEY_INTERVAL - 1
        if num_interms < 1:
            raise ValueError("TRAIN.KEY_INTERVAL must be >= 2")
        data = mx.sym.Variable(name="data")
        data_ref = mx.sym.Variable(name="data_ref")
        eq_flag = mx.sym.Variable(name="eq_flag")
        seg_cls_gt = mx.symbol.Variable(name='label')
        data_ref_split = mx.sym.split(data_ref, num_outputs=num_interms, axis=0)

        if self.version == '101':
            # key frame and current frame through ONE ResNet-101 batch (accel_101.py:46-49)
            feat_concat = self.get_resnet_dcn(mx.sym.Concat(*[data_ref_split[0], data], dim=0))
            feat_split = mx.sym.split(feat_concat, num_outputs=2, axis=0)
            conv_feat, feat_curr = feat_split[0], feat_split[1]
        else:
            conv_feat = self.get_resnet_dcn(data_ref_split[0])
        # frame pairs (next, prev) = (ref1, ref0), (ref2, ref1), ..., (data, ref_last): accel_18.py:46-48
        data_next = mx.sym.Concat(*([data_ref_split[i] for i in range(1, num_interms)] + [data]), dim=0)
        data_prev = mx.sym.Concat(*[data_ref_split[i] for i in range(num_interms)], dim=0)
        flow, scale_map = self.get_flownet(data_next, data_prev)
        flow_grid = mx.sym.GridGenerator(data=flow, transform_type='warp', name='flow_grid')
        flow_grid_split = mx.sym.split(flow_grid, num_outputs=num_interms, axis=0)
        for idx in range(num_interms):
            conv_feat = mx.sym.BilinearSampler(data=conv_feat, grid=flow_grid_split[idx], name='warping_feat')

        if self.version == '101':
            stacked_in = mx.sym.Concat(*[conv_feat, feat_curr], dim=1)
            feat_fuse = self._correction(stacked_in, 2048)
            logits = self._task_head(feat_fuse, data, num_classes)
        else:
            croped_score = self._task_head(conv_feat, data, num_classes)
            feat_curr = self._r_branch_features(data)
            curr_croped_score = self._task_head(feat_curr, data, num_classes, self.branch_prefix)
            stacked_in = mx.sym.Concat(*[croped_score, curr_croped_score], dim=1)
            logits = self._correction(stacked_in, num_classes)
        softmax = mx.symbol.SoftmaxOutput(data=logits, label=seg_cls_gt, normalization='valid', multi_output=True,
                                          use_ignore=True, ignore_label=255, name="softmax")
        group = mx.sym.Group([softmax, data_ref, eq_flag])
        self.sym = group
        return group

    def get_batch_test_symbol(self, cfg):
        raise NotImplementedError("R-FCN detection leftover; needs MultiProposal/PSROIPooling (out of scope)")

    def init_weight(self, cfg, arg_params, aux_params, rng=None):
        """Initialisation of the parameters a pretrained DeepLab / DFF checkpoint lacks (run before fine-tuning; needs
        infer_shape first).  accel_18.py:321-323 / accel_34.py:321-323: corr_weight ~ N(0, 0.01), corr_bias = 0;
        accel_50.py:310-318: the same plus the `curr_*` head copied from the `50_*` head of the checkpoint;
        accel_101.py:275-289: corr = [0 | I] -- the fusion conv starts as "pass the current frame's feature through" --
        and the `curr_*` head copied from the key head."""
        import numpy as np
        rng = rng or np.random.default_rng()
        shape = tuple(self.arg_shape_dict['corr_weight'])
        if self.version == '101':
            w = np.zeros(shape, np.float32)
            for i in range(shape[0]):
                w[i, shape[0] + i] = 1.0
            arg_params['corr_weight'] = w
        else:
            arg_params['corr_weight'] = rng.normal(0, 0.01, shape).astype(np.float32)
        arg_params['corr_bias'] = np.zeros(tuple(self.arg_shape_dict['corr_bias']), np.float32)
        src = {'101': '', '50': '50_'}.get(self.version)
        if src is not None:
            for nm in ('fc6_weight', 'fc6_bias', 'score_weight', 'score_bias', 'upsampling_weight'):
                if src + nm in arg_params:
                    arg_params['curr_' + nm] = arg_params[src + nm]


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
