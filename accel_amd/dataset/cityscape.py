"""Cityscapes file layout, palette PNG writer and confusion-matrix mIoU:
counterpart of lib/dataset/cityscape.py (CityScape :17-58 paths, :123-168
palette, :172-241 evaluation, :243-270 result writer).  PIL + numpy only.

Layout (dataset_path = .../cityscapes):
    <main>/<sub>/<city>/<city>_<seq>_<frame>_<main>.png      e.g. leftImg8bit/val/frankfurt/..._leftImg8bit.png
    gtFine/<sub>/<city>/<city>_<seq>_<frame>_gtFine_trainIds.png
    results/<city>/<city>_<seq>_<frame>.png                   palette PNGs written by write_segmentation_result
"""
import itertools
import os

import numpy as np
from PIL import Image

# Cityscapes label colours by regular label id (7..33 are the ones trainIds map onto)
_LABEL_COLOURS = {
    7: (128, 64, 128), 8: (244, 35, 232), 11: (70, 70, 70), 12: (102, 102, 156), 13: (190, 153, 153),
    17: (153, 153, 153), 19: (250, 170, 30), 20: (220, 220, 0), 21: (107, 142, 35), 22: (152, 251, 152),
    23: (70, 130, 180), 24: (220, 20, 60), 25: (255, 0, 0), 26: (0, 0, 142), 27: (0, 0, 70),
    28: (0, 60, 100), 31: (0, 80, 100), 32: (0, 0, 230), 33: (119, 11, 32)}
TRAIN2REGULAR = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]


def getpallete(num_cls=256):
    """Flat RGB palette: trainId i -> colour of regular id TRAIN2REGULAR[i]; the rest black
    (cityscape.py:123-168, demo.py:58-105)."""
    pal = np.zeros((num_cls, 3), np.uint8)
    for i, rid in enumerate(TRAIN2REGULAR):
        pal[i] = _LABEL_COLOURS[rid]
    return pal.reshape(-1)


def confusion_matrix(gt_label, pred_label, class_num):
    """rows = ground truth, columns = prediction (cityscape.py:185-203)"""
    idx = gt_label.astype(np.int64) * class_num + pred_label.astype(np.int64)
    return np.bincount(idx, minlength=class_num * class_num)[:class_num * class_num].reshape(class_num, class_num).astype(np.float64)


def _nearest_resize(a, h, w):
    if a.shape == (h, w):
        return a
    ys = np.minimum((np.arange(h) * a.shape[0] / float(h)).astype(int), a.shape[0] - 1)
    xs = np.minimum((np.arange(w) * a.shape[1] / float(w)).astype(int), a.shape[1] - 1)
    return a[ys][:, xs]


class CityScape(object):
    def __init__(self, image_set, root_path, dataset_path, result_path=None):
        main, sub = image_set.split('_', 1)            # 'leftImg8bit_val' -> ('leftImg8bit', 'val')
        self.name = 'cityscape_' + image_set
        self.image_set = image_set
        self.image_set_main_folder, self.image_set_sub_folder = main, sub
        self.root_path = root_path
        self.data_path = dataset_path
        self.result_path = result_path if result_path else dataset_path
        self.num_classes = 19
        self.image_set_index = self.load_image_set_index()
        self.num_images = len(self.image_set_index)

    def load_image_set_index(self):
        folder = os.path.join(self.data_path, self.image_set_main_folder, self.image_set_sub_folder)
        names = itertools.chain.from_iterable(f for _, _, f in sorted(os.walk(folder)))
        index = []
        for n in sorted(names):
            parts = n.split('_')
            if n.endswith('.png') and parts[-1] != 'flip.png':
                index.append('_'.join(parts[:-1]))
        return index

    def image_path_from_index(self, index):
        p = os.path.join(self.data_path, self.image_set_main_folder, self.image_set_sub_folder, index.split('_')[0],
                         index + '_' + self.image_set_main_folder + '.png')
        assert os.path.exists(p), 'Path does not exist: {}'.format(p)
        return p

    def annotation_path_from_index(self, index):
        p = os.path.join(self.data_path, 'gtFine', self.image_set_sub_folder, index.split('_')[0],
                         index + '_gtFine_trainIds.png')
        assert os.path.exists(p), 'Path does not exist: {}'.format(p)
        return p

    def load_segdb_from_index(self, index):
        image = self.image_path_from_index(index)
        with Image.open(image) as im:
            w, h = im.size
        return {'image': image, 'height': h, 'width': w, 'seg_cls_path': self.annotation_path_from_index(index),
                'flipped': False}

    def gt_segdb(self):
        return [self.load_segdb_from_index(i) for i in self.image_set_index]

    def getpallete(self, num_cls):
        return getpallete(num_cls)

    get_confusion_matrix = staticmethod(confusion_matrix)

    def _result_path(self, seg_cls_path):
        folder, fname = os.path.split(seg_cls_path)
        return os.path.join(self.result_path, 'results', os.path.basename(folder),
                            fname[:-len('_gtFine_trainIds.png')] + '.png')

    def write_segmentation_result(self, segmentation_results):
        pal = getpallete(256)
        for i, index in enumerate(self.image_set_index):
            out = self._result_path(self.annotation_path_from_index(index))
            os.makedirs(os.path.dirname(out), exist_ok=True)
            img = Image.fromarray(np.uint8(np.squeeze(segmentation_results[i])))
            img.putpalette(pal)
            img.save(out)

    def _py_evaluate_segmentation(self):
        cm = np.zeros((self.num_classes, self.num_classes))
        for index in self.image_set_index:
            gt_path = self.annotation_path_from_index(index)
            gt = np.array(Image.open(gt_path)).astype(np.int64)
            pred = np.array(Image.open(self._result_path(gt_path))).astype(np.int64)
            pred = _nearest_resize(pred, gt.shape[0], gt.shape[1])
            keep = gt != 255
            cm += confusion_matrix(gt[keep], pred[keep], self.num_classes)
        pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
        iu = tp / np.maximum(1.0, pos + res - tp)
        return {'meanIU': iu.mean(), 'IU_array': iu}

    def evaluate_segmentations(self, pred_segmentations=None):
        if pred_segmentations is not None:
            self.write_segmentation_result(pred_segmentations)
        return self._py_evaluate_segmentation()
