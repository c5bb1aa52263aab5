"""Accel inference + mIoU + timing harness: the counterpart of
dff_deeplab/demo.py (the only working segmentation harness of the reference,
SURVEY.md F2), on the HIP runtime.

    python -m accel_amd.demo --version 18 --interval 5 --num_ex 10 [--avg]
                             [--data DIR --labels DIR | --synthetic HxW]
                             [--params accel-18-0000.params flownet-0000.params]

Reproduced behaviour (demo.py:107-284): frame selection (`lb_pos = 19`,
`offset = interval-1` or `i % interval` with --avg), key/non-key schedule
(`idx % interval == 0`), `data_key` = PREVIOUS frame, feature feedback,
`correction_output` vs `croped_score_output` (Accel-101), 2 warm-up frames,
tic/toc around forward + argmax + label-map fetch, fast_hist/per_class_iu mIoU.
Without Cityscapes on disk the clip source is synthetic (`--synthetic`).
"""
import argparse
import glob
import os

import numpy as np

from . import mx
from .config.config import config, update_config
from .core.tester import Predictor, im_segment
from .utils.image import resize, transform
from .utils.tictoc import tic, toc


def fast_hist(pred, label, n):
    """demo.py:50-53"""
    k = (label >= 0) & (label < n)
    return np.bincount(n * label[k].astype(int) + pred[k], minlength=n ** 2).reshape(n, n)


def per_class_iu(hist):
    """demo.py:55-56"""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))


def select_frames(image_names, num_ex, interv, avg_acc, snip_len=30, lb_pos=19):
    """demo.py:152-163: per 30-frame snippet keep `interv` frames positioned so
    that the labelled frame (index 19) is `offset` frames after the key frame."""
    out = []
    for i in range(num_ex):
        snip_pos = i * snip_len
        offset = i % interv if avg_acc else interv - 1
        start_pos = lb_pos - offset
        out.extend(image_names[snip_pos + start_pos: snip_pos + start_pos + interv])
    return out


def label_key(path):
    """(city, sequence, frame) of a Cityscapes file name `<city>_<seq>_<frame>_<suffix>.png` -- the city is part of the
    key: lindau_0000NN_000019 and munster_0000NN_000019 both exist in val for NN = 0..58 (the reference walks the label
    list in order with lb_idx instead, demo.py:140-150,262-266)."""
    c = os.path.basename(path).split('_')
    return tuple(c[:3]) if len(c) > 3 else None


def get_symbols(version, cfg):
    from . import symbols
    name = "accel_" + str(version)
    inst = getattr(getattr(symbols, name), name)()
    return inst, inst.get_key_test_symbol(cfg), inst.get_cur_test_symbol(cfg)


def build_batches(frames_bgr, cfg, pinned=False):
    """demo.py:165-190: list of [data, data_key, feat_key] arrays per frame.

    One array per frame serves as this frame's `data` and as the next frame's `data_key` (the reference builds two
    NDArrays from the same image, demo.py:176-181): arrays are immutable, so the Predictor recognises the object it
    uploaded one call earlier and copies the image inside HBM instead of sending it over PCIe twice.
    pinned=True places the images in page-locked memory (mx.cpu_pinned()), the source of overlapped uploads."""
    data, prev = [], None
    ctx = mx.cpu_pinned() if pinned else None
    zero_feat = mx.nd.array(np.zeros((1, cfg.network.DFF_FEAT_DIM, 1, 1)))
    for im in frames_bgr:
        target_size, max_size = cfg.SCALES[0][0], cfg.SCALES[0][1]
        im, _ = resize(im, target_size, max_size, stride=cfg.network.IMAGE_STRIDE)
        cur = mx.nd.array(transform(im, cfg.network.PIXEL_MEANS), ctx=ctx)
        if prev is None:
            prev = cur
        data.append([cur, prev, zero_feat])
        prev = cur
    return data


class ClipRunner(object):
    """The demo.py hot loop as an object: a key and a cur Predictor sharing one device."""

    data_names = ['data', 'data_key', 'feat_key']

    _tags = 0

    def __init__(self, version, cfg, arg_params, aux_params, frame_hw, context=None, model=None, batch=1, share_models=False):
        """`model`: bind both predictors on this runtime.Model (the bench does).  Otherwise the runner's two predictors
        share models with each other only -- two runners interleaving clips of the same size do not see each other's
        propagated feature -- unless share_models=True (one model per device, size and parameter content, as for
        predictors built directly)."""
        self.version = str(version)
        ClipRunner._tags += 1
        self.owner = None if (share_models or model is not None) else "cliprunner-%d" % ClipRunner._tags
        self.cfg = cfg
        H, W = frame_hw
        _, key_sym, cur_sym = get_symbols(version, cfg)
        ctx = context or [mx.gpu(0)]
        B = int(batch)      # frames per call: one frame of each of B independent clips (throughput mode)
        max_shape = [[('data', (B, 3, H, W)), ('data_key', (B, 3, H, W))]]
        provide = [[('data', (B, 3, H, W)), ('data_key', (B, 3, H, W)), ('feat_key', (B, 2048, 1, 1))]]
        self.key_predictor = Predictor(key_sym, self.data_names, [], context=ctx, max_data_shapes=max_shape,
                                       provide_data=provide, provide_label=[None],
                                       arg_params=arg_params, aux_params=aux_params, model=model, owner=self.owner)
        self.cur_predictor = Predictor(cur_sym, self.data_names, [], context=ctx, max_data_shapes=max_shape,
                                       provide_data=provide, provide_label=[None],
                                       arg_params=arg_params, aux_params=aux_params, model=model, owner=self.owner)
        self.feat = None
        self.output_key = 'croped_score_output' if self.version in ('101', 'dff') else 'correction_output'

    def close(self):
        """release the models this runner owns (no-op for shared / explicit models)"""
        if self.owner is not None:
            from .core import tester
            tester.release_models(self.owner)

    def prefetch(self, arrays):
        """start the upload of the NEXT frame's image (page-locked arrays only) beside the running forward"""
        return self.key_predictor.prefetch(arrays[0])

    def step(self, idx, arrays, interval):
        """One frame (demo.py:235-245).  Returns (logits handle, label-map handle)."""
        batch = mx.io.DataBatch(data=[list(arrays)], label=[], pad=0, index=idx,
                                provide_data=[[(k, v.shape) for k, v in zip(self.data_names, arrays)]],
                                provide_label=[None])
        if idx % interval == 0:
            output_all, self.feat = im_segment(self.key_predictor, batch)
            logits = output_all[0]['croped_score_output']
        else:
            batch.data[0][-1] = self.feat
            batch.provide_data[0][-1] = ('feat_key', self.feat.shape)
            output_all, self.feat = im_segment(self.cur_predictor, batch)
            logits = output_all[0][self.output_key]
        return logits, mx.nd.argmax(logits, axis=1)


def run_clip(version, cfg, arg_params, aux_params, frames_bgr, interval, want_logits=True):
    """Runs the demo schedule over `frames_bgr`; returns per-frame (logits, labels) numpy."""
    data = build_batches(frames_bgr, cfg)
    H, W = data[0][0].shape[2:]
    runner = ClipRunner(version, cfg, arg_params, aux_params, (H, W))
    outs = []
    for idx, arrays in enumerate(data):
        logits, labels = runner.step(idx, arrays, interval)
        lab = np.uint8(np.squeeze(labels.asnumpy()))
        outs.append((logits.asnumpy() if want_logits else None, lab))
    return outs


def main(argv=None):
    ap = argparse.ArgumentParser(description='Accel demo (MI355X)')
    ap.add_argument('--version', default='18')
    ap.add_argument('--interval', type=int, default=5)
    ap.add_argument('--num_ex', type=int, default=10)
    ap.add_argument('--avg', dest='avg_acc', action='store_true')
    ap.add_argument('--cfg', default=None, help='experiments/dff_deeplab/cfgs/dff_deeplab_vid_demo.yaml')
    ap.add_argument('--data', default='', help='Cityscapes root (leftImg8bit_sequence/, gtFine/)')
    ap.add_argument('--synthetic', default='1024x2048')
    ap.add_argument('--params', nargs='*', default=[], help='MXNet .params checkpoints, merged in order')
    ap.add_argument('--out', default='', help='directory for palette PNGs seg_<frame>.png (demo.py:252-257)')
    ap.add_argument('--pageable', action='store_true',
                    help='frames in pageable host memory, uploaded synchronously inside each forward (the reference\'s '
                         'loop one to one); default: page-locked frames, the next frame\'s upload overlaps the current forward')
    args = ap.parse_args(argv)
    version, interv, num_ex = str(args.version), args.interval, args.num_ex
    if version not in ['18', '34', '50', '101', 'dff']:
        raise ValueError("Invalid Accel version '%s' - must be one of Accel-{18,34,50,101} (or 'dff')" % version)
    if interv < 1:
        raise ValueError("Invalid interval %d - must be >=1" % interv)
    if num_ex < 1:
        raise ValueError("Invalid num_ex %d - must be >=1" % num_ex)
    if args.cfg:
        update_config(args.cfg)
    num_classes = config.dataset.NUM_CLASSES

    labels = {}
    if args.data:
        from PIL import Image
        names, label_files = [], []
        for city in ('frankfurt', 'lindau', 'munster'):
            names += sorted(glob.glob(os.path.join(args.data, 'leftImg8bit_sequence/val', city, '*.png')))
            label_files += sorted(glob.glob(os.path.join(args.data, 'gtFine/val', city, '*trainIds.png')))
        names = select_frames(names[:30 * num_ex], num_ex, interv, args.avg_acc)
        frames = [np.asarray(Image.open(n).convert('RGB'))[:, :, ::-1] for n in names]
        for lf in label_files:
            labels[label_key(lf)] = lf
    else:
        from .utils import synth
        H, W = [int(v) for v in args.synthetic.split('x')]
        config.SCALES[0] = (H, W)
        frames, names = [], []
        for i in range(num_ex):
            clip = synth.make_clip(H, W, interv, seed=20260929 + i)
            frames += clip
            names += ['synthetic_%06d_%06d_leftImg8bit.png' % (i, t) for t in range(interv)]

    from .utils import load_model, synth
    H, W = frames[0].shape[:2]
    if args.params:
        arg_params, aux_params = {}, {}
        for prefix in args.params:
            a, x = load_model.load_param_file(prefix, process=True)
            arg_params.update(a)
            aux_params.update(x)
    else:
        print('no --params given: seeded random weights (throughput is valid, mIoU is meaningless)')
        arg_params, aux_params = synth.model_params(version, H, W, config)

    data = build_batches(frames, config, pinned=not args.pageable)
    runner = ClipRunner(version, config, arg_params, aux_params, (H, W))
    for j in range(min(2, len(data))):       # warm up (demo.py:207-220)
        runner.step(j, data[j], interv)[1].asnumpy()
    print("warmup done")
    time_sum, count = 0.0, 0
    hist = np.zeros((num_classes, num_classes))
    for idx, arrays in enumerate(data):
        tic()
        _, lab = runner.step(idx, arrays, interv)
        if idx + 1 < len(data) and not args.pageable:
            runner.prefetch(data[idx + 1])        # next frame starts crossing PCIe while this one computes
        pred = np.uint8(np.squeeze(lab.asnumpy()))
        elapsed = toc()
        time_sum += elapsed
        count += 1
        print('testing {} {:.4f}s [{:.4f}s]'.format(names[idx], elapsed, time_sum / count))
        if args.out:
            from PIL import Image
            from .dataset.cityscape import getpallete
            os.makedirs(args.out, exist_ok=True)
            seg = Image.fromarray(pred)
            seg.putpalette(getpallete(256))
            seg.save(os.path.join(args.out, 'seg_' + os.path.basename(names[idx])))
        lf = labels.get(label_key(names[idx]))
        if lf is not None:
            from PIL import Image
            label = np.asarray(Image.open(lf))
            curr_hist = fast_hist(pred.flatten(), label.flatten(), num_classes)
            hist += curr_hist
            print('mIoU {mIoU:.3f}'.format(mIoU=round(np.nanmean(per_class_iu(curr_hist)) * 100, 2)))
            print('(cum) mIoU {mIoU:.3f}'.format(mIoU=round(np.nanmean(per_class_iu(hist)) * 100, 2)))
    if hist.sum() > 0:
        ious = per_class_iu(hist) * 100
        print(' '.join('{:.03f}'.format(i) for i in ious))
        print('===> final mIoU {mIoU:.3f}'.format(mIoU=round(np.nanmean(ious), 2)))
    print('{:.2f} frames/s'.format(count / time_sum))
    print('done')


if __name__ == '__main__':
    main()
