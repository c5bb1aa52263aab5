"""Host-side frame preprocessing: counterparts of lib/utils/image.py:194-235."""
import numpy as np


def _resize_bilinear(im, out_h, out_w, scale_y=None, scale_x=None):
    """cv2.resize(..., INTER_LINEAR) geometry: half-pixel centres, source coordinate (dst + 0.5) * scale - 0.5 clamped to the
    image, where scale = 1 / fx when the caller gave fx (cv2 keeps the REQUESTED factor, not in / out, when dsize is derived
    from it -- lib/utils/image.py:209 calls it that way) and in / out otherwise.  Arithmetic in float64 with one final
    rounding; cv2's uint8 path uses 11-bit fixed-point weights and may differ by one grey level.  (The BASELINE path never
    gets here: 1024x2048 frames at SCALES (1024, 2048) have scale exactly 1.)"""
    h, w = im.shape[:2]
    if (h, w) == (out_h, out_w):
        return im
    sy = float(h) / out_h if scale_y is None else scale_y
    sx = float(w) / out_w if scale_x is None else scale_x
    ys = np.clip((np.arange(out_h) + 0.5) * sy - 0.5, 0, h - 1)
    xs = np.clip((np.arange(out_w) + 0.5) * sx - 0.5, 0, w - 1)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = im.astype(np.float64)
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    if im.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out


def resize(im, target_size, max_size, stride=0):
    """Contract of lib/utils/image.py:194-222: the SHORT side goes to `target_size` unless that would push the long side past
    `max_size` (then the long side goes to `max_size`); with `stride` > 0 the result is zero-padded at the bottom / right to the
    next multiple of it (as float64, like the reference's pad buffer).  Returns (image, scale).  1024x2048 Cityscapes frames at
    SCALES (1024, 2048): scale exactly 1, nothing is resampled."""
    rows, cols = im.shape[:2]
    short, long_ = (rows, cols) if rows <= cols else (cols, rows)
    scale = float(target_size) / float(short)
    if np.round(scale * long_) > max_size:
        scale = float(max_size) / float(long_)
    if scale != 1.0:
        im = _resize_bilinear(im, int(round(rows * scale)), int(round(cols * scale)), 1.0 / scale, 1.0 / scale)
    if stride == 0:
        return im, scale
    up = lambda n: -(-n // stride) * stride         # next multiple of the stride
    canvas = np.zeros((up(im.shape[0]), up(im.shape[1]), im.shape[2]))
    canvas[:im.shape[0], :im.shape[1]] = im
    return canvas, scale


def transform(im, pixel_means):
    """Contract of lib/utils/image.py:224-235: a BGR H x W x 3 frame becomes the 1 x 3 x H x W RGB tensor with the per-channel
    mean removed (`pixel_means` is given in B, G, R order like the frame).  float64 like the reference; arrays become fp32
    when they are handed to the predictor (demo.py:186)."""
    centred = np.asarray(im, np.float64) - np.asarray(pixel_means, np.float64).reshape(1, 1, 3)
    return np.ascontiguousarray(centred[:, :, ::-1].transpose(2, 0, 1))[None]
