"""Frame-by-frame DeepLab-v2 ResNet-101-DCN baseline (reference
deeplab/symbols/resnet_v1_101_deeplab_dcn.py: get_resnet_conv :24-748,
get_test_symbol :786-821, get_symbol :823-836).  It is the Accel key-frame
graph with a SoftmaxOutput on top and `data` as its only input, so it runs on
the same kernels (README "Main Results": DeepLab R101 row)."""
from .. import mx
from .accel_base import accel_base


class resnet_v1_101_deeplab_dcn(accel_base):
    version = 'deeplab'

    def get_resnet_conv(self, data):
        return self.get_resnet_dcn(data)

    def get_test_symbol(self, num_classes):
        data = mx.symbol.Variable(name="data")
        conv_feat = self.get_resnet_conv(data)
        croped_score = self._task_head(conv_feat, data, num_classes)
        softmax = mx.symbol.SoftmaxOutput(data=croped_score, normalization='valid', multi_output=True,
                                          use_ignore=True, ignore_label=255, name="softmax")
        return softmax

    def get_symbol(self, cfg, is_train=True):
        if is_train:
            raise NotImplementedError("training graphs are outside the inference hot path")
        self.sym = self.get_test_symbol(cfg.dataset.NUM_CLASSES)
        return self.sym
