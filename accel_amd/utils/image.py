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
    """Scale the short side to target_size, capped so the long side <= max_size;
    optional zero padding to a multiple of `stride` (image.py:194-222).  For
    1024x2048 Cityscapes frames at SCALES (1024, 2048) the scale is exactly 1."""
    im_shape = im.shape
    im_size_min = np.min(im_shape[0:2])
    im_size_max = np.max(im_shape[0:2])
    im_scale = float(target_size) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    if im_scale != 1.0:
        im = _resize_bilinear(im, int(round(im_shape[0] * im_scale)), int(round(im_shape[1] * im_scale)), 1.0 / im_scale, 1.0 / im_scale)
    if stride == 0:
        return im, im_scale
    im_height = int(np.ceil(im.shape[0] / float(stride)) * stride)
    im_width = int(np.ceil(im.shape[1] / float(stride)) * stride)
    padded_im = np.zeros((im_height, im_width, im.shape[2]))
    padded_im[:im.shape[0], :im.shape[1], :] = im
    return padded_im, im_scale


def transform(im, pixel_means):
    """BGR HxWx3 -> 1x3xHxW RGB minus means; pixel_means is [B, G, R] indexed 2-i
    (image.py:224-235).  float64 like the reference; arrays become fp32 when they
    are handed to the predictor (demo.py:186)."""
    im_tensor = np.zeros((1, 3, im.shape[0], im.shape[1]))
    for i in range(3):
        im_tensor[0, i, :, :] = im[:, :, 2 - i] - pixel_means[2 - i]
    return im_tensor
