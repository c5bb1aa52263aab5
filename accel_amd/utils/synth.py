"""Seeded synthetic weights and clips (no checkpoints or Cityscapes data exist
offline).

Weights are generated per tensor from seed = crc32(name), so any subset of a
model regenerates identically and nothing but the seed rule has to be stored
with a golden fixture.  The distributions keep activations O(1) through the
100+ layers so that the absolute logit tolerance of the parity tests
(1e-3, BASELINE.json north_star) is meaningful:
  conv / deconv weights  N(0, sqrt(2 / fan_in))      (He)
  biases                 N(0, 0.01)
  BN gamma U(0.5, 1.5) (x0.25 on the last BN of a residual branch, so the
       residual sum does not double the variance 33 times), beta N(0, 0.1),
       moving_mean N(0, 0.1), moving_var U(0.5, 1.5)
  *upsampling_weight     bilinear kernel (frozen in the reference, lr_mult 0;
       accel_18.py:153)
  DCN offset convs       N(0, offset_std) weights -> offsets of about a pixel: the offset convs see
       activations of rms ~7 (ResNet-18 conv5) to ~14 (ResNet-101 res5) over a fan-in of 4608, so offset_std =
       0.0016 gives offsets of std 0.7-1.5 px.  (Round 1 used 0.02: mean |offset| 12 px, max 53 -- most taps of a
       stride-16 map then land OUTSIDE the image, where the DCN-v1 border rule is discontinuous (zero for h < 0, the
       pixel value at h = 0): any rounding difference then flips isolated feature pixels by several units, see
       DESIGN.md "numerics".  Large offsets stay covered by the operator-level stress tests.)
  FlowNet flow predictors scaled so |flow| is a few feature pixels
  Accel-101 corr_weight (2048 x 4096 feature fusion): 0.5 [I | I] + 0.25 x He noise, so that non-key frames keep several classes
"""
import zlib

import numpy as np


def _rng(name, salt=0):
    return np.random.default_rng((zlib.crc32(name.encode()) + salt) & 0xFFFFFFFF)


def bilinear_kernel(channels, k):
    """The standard FCN bilinear upsampling filter, shape (channels, 1, k, k)."""
    f = (k + 1) // 2
    c = f - 1 if k % 2 == 1 else f - 0.5
    og = np.ogrid[:k, :k]
    filt = (1 - abs(og[0] - c) / f) * (1 - abs(og[1] - c) / f)
    return np.tile(filt.astype(np.float32)[None, None], (channels, 1, 1, 1))


def _is_residual_tail_bn(name):
    # last BN of a bottleneck / basic residual branch in the post-activation nets
    base = name.rsplit("_", 1)[0]
    return base.endswith("_branch2c") or (base.startswith(("18_bn5", "34_bn5")) and base.endswith("_branch2b"))


def make_param(name, shape, offset_std=0.0016, flow_gain=1.0, salt=0):
    shape = tuple(int(s) for s in shape)
    r = _rng(name, salt)
    if name.endswith("upsampling_weight") and len(shape) == 4 and shape[1] == 1:
        return bilinear_kernel(shape[0], shape[2])
    if name.endswith("_gamma"):
        g = r.uniform(0.5, 1.5, shape)
        if _is_residual_tail_bn(name):
            g *= 0.25
        return g.astype(np.float32)
    if name.endswith("_beta") or name.endswith("_moving_mean"):
        return r.normal(0, 0.1, shape).astype(np.float32)
    if name.endswith("_moving_var"):
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if name.endswith("_bias"):
        return r.normal(0, 0.01, shape).astype(np.float32)
    if name == "corr_weight" and len(shape) == 4 and shape[0] >= 1024 and shape[1] == 2 * shape[0] and shape[2:] == (1, 1):
        # Accel-101's feature fusion (accel_101.py:171-176: 1x1 conv over concat(warped key feature, current feature), 4096 -> 2048):
        # an averaging fusion 0.5 [I | I] plus a quarter of the He noise.  A pure He draw gives every fused channel a DC term
        # (4096 all-positive inputs) that swamps the spatial signal, and the non-key label map of the seeded model is then ONE
        # class at every pixel -- "labels identical" would prove nothing there (round-3 review); with this rule the non-key
        # frames carry 5-6 classes like the key frame.
        eye = np.concatenate([np.eye(shape[0]), np.eye(shape[0])], axis=1).reshape(shape)
        return (0.5 * eye + 0.25 * r.normal(0, np.sqrt(2.0 / shape[1]), shape)).astype(np.float32)
    if name.endswith("_weight"):
        if "_offset_" in name:
            return r.normal(0, offset_std, shape).astype(np.float32)
        if "feat_upsampling" in name:          # Deconvolution (Cin, Cout, 4, 4): 4 taps hit each output
            fan_in = shape[0] * 4
        elif name.startswith(("deconv", "upsample_flow")):
            fan_in = shape[0] * 4
        else:
            fan_in = int(np.prod(shape[1:]))
        w = r.normal(0, np.sqrt(2.0 / fan_in), shape)
        if name.startswith("Convolution") and shape[0] == 2:   # FlowNet flow predictors
            w *= flow_gain
        if len(shape) == 4 and shape[1] == 3 and shape[2] == 7:    # stems fed with raw +-128 pixel values
            w *= 1.0 / 64.0
        if name.startswith("stage") or "_stage" in name:
            if name.endswith("_conv2_weight"):                 # pre-act residual branch tail
                w *= 0.5
        return w.astype(np.float32)
    raise ValueError("no generator rule for parameter %r" % name)


def make_params(arg_shapes, aux_shapes, data_names=("data", "data_key", "feat_key"), **kw):
    arg = {k: make_param(k, s, **kw) for k, s in arg_shapes.items() if k not in data_names}
    aux = {k: make_param(k, s, **kw) for k, s in aux_shapes.items()}
    return arg, aux


def model_params(version, H, W, cfg=None, **kw):
    """arg/aux dicts covering BOTH test graphs of Accel-<version> at HxW."""
    from ..config.config import config as default_cfg
    from .. import symbols
    cfg = cfg or default_cfg
    name = "accel_" + str(version)
    inst = getattr(getattr(symbols, name), name)()     # '18' | '34' | '50' | '101' | 'dff'
    arg, aux = {}, {}
    shp = {"data": (1, 3, H, W), "data_key": (1, 3, H, W)}
    for getter, feat in ((inst.get_key_test_symbol, (1, 2048, 1, 1)),
                         (inst.get_cur_test_symbol, (1, 2048, H // 16, W // 16))):
        getter(cfg)
        inst.infer_shape(dict(shp, feat_key=feat))
        a, x = make_params(inst.arg_shape_dict, inst.aux_shape_dict, **kw)
        arg.update(a)
        aux.update(x)
    return arg, aux


# ---------------------------------------------------------------------------
# frames
# ---------------------------------------------------------------------------
def _upsample_bilinear(a, H, W):
    h, w = a.shape[:2]
    ys = (np.arange(H) + 0.5) * h / H - 0.5
    xs = (np.arange(W) + 0.5) * w / W - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    y1 = np.clip(y0 + 1, 0, h - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    fy = np.clip(ys - y0, 0, 1)[:, None, None]
    fx = np.clip(xs - x0, 0, 1)[None, :, None]
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def base_texture(H, W, seed=20260929):
    rng = np.random.default_rng(seed)
    img = np.zeros((H, W, 3), np.float64)
    for o in range(5):
        d = 2 ** (7 - o)
        h, w = max(H // d, 2), max(W // d, 2)
        img += _upsample_bilinear(rng.uniform(0, 255, (h, w, 3)), H, W) * 2.0 ** -(4 - o)
    img -= img.min()
    img *= 255.0 / max(img.max(), 1e-9)
    return img


def make_clip(H, W, n_frames, seed=20260929, dx=2, dy=1):
    """uint8 BGR frames (H, W, 3): a textured image translating by (dx, dy) px
    per frame (wrap-around) plus N(0, 2) noise -- smooth, non-trivial flow."""
    base = base_texture(H, W, seed)
    rng = np.random.default_rng(seed + 1)
    frames = []
    for t in range(n_frames):
        f = np.roll(base, (dy * t, dx * t), axis=(0, 1)) + rng.normal(0, 2, base.shape)
        frames.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    return frames
