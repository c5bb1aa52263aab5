"""Global experiment config with the reference's defaults and YAML overlay
rules (dff_deeplab/config/config.py:12-116): top-level keys must pre-exist,
sub-keys are added freely, SCALES becomes [(h, w)], PIXEL_MEANS an array.
`experiments/dff_deeplab/cfgs/*.yaml` of the reference load unchanged."""
import numpy as np
import yaml


class edict(dict):
    """Attribute-access dict (the reference uses easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        dict.__init__(self)
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, edict):
            v = edict(v)
        dict.__setitem__(self, k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _defaults():
    c = edict()
    c.MXNET_VERSION = ''
    c.output_path = ''
    c.symbol = ''
    c.gpus = ''
    c.CLASS_AGNOSTIC = True
    c.SCALES = [(360, 600)]
    c.default = edict(frequent=20, kvstore='device')
    c.network = edict(pretrained='', pretrained_base='', pretrained_ec='', pretrained_epoch=0,
                      PIXEL_MEANS=np.array([0, 0, 0]), IMAGE_STRIDE=0, FIXED_PARAMS=['gamma', 'beta'],
                      DFF_FEAT_DIM=2048)
    c.dataset = edict(dataset='CityScape', image_set='leftImg8bit_train', test_image_set='leftImg8bit_val',
                      root_path='../data', dataset_path='../data/cityscapes', NUM_CLASSES=19,
                      annotation_prefix='gtFine')
    c.TRAIN = edict(lr=0, lr_step='', lr_factor=0.1, warmup=False, warmup_lr=0, warmup_step=0, momentum=0.9,
                    wd=0.0005, begin_epoch=0, end_epoch=0, model_prefix='', RESUME=False, FLIP=True,
                    SHUFFLE=True, ENABLE_OHEM=False, BATCH_IMAGES=1, END2END=False, ASPECT_GROUPING=True,
                    MIN_OFFSET=-4, MAX_OFFSET=0, KEY_INTERVAL=5)
    c.TEST = edict(BATCH_IMAGES=1, KEY_FRAME_INTERVAL=5, max_per_image=300, test_epoch=0)
    return c


config = _defaults()


def reset_config():
    config.clear()
    for k, v in _defaults().items():
        config[k] = v


def update_config(config_file):
    with open(config_file) as f:
        exp_config = edict(yaml.safe_load(f))
    for k, v in exp_config.items():
        if k not in config:
            raise ValueError("key must exist in config.py")
        if isinstance(v, dict):
            if k == 'TRAIN' and 'BBOX_WEIGHTS' in v:
                v['BBOX_WEIGHTS'] = np.array(v['BBOX_WEIGHTS'])
            elif k == 'network' and 'PIXEL_MEANS' in v:
                v['PIXEL_MEANS'] = np.array(v['PIXEL_MEANS'])
            for vk, vv in v.items():
                config[k][vk] = vv
        elif k == 'SCALES':
            config[k][0] = tuple(v)
        else:
            config[k] = v
