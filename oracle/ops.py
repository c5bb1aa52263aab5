"""ctypes/numpy front-end of the CPU oracle (oracle/accel_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of accel_oracle.c.  Nothing under
accel_amd/ may import this module; tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() use it as the checker.  PARITY UNPINNED (no reference
golden vectors exist).

Every function takes and returns fp32 NCHW numpy arrays, like the MXNet
operators the reference's symbol files call.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "accel_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        # many-core hosts: the oracle's tensors are small, 256 OpenMP threads only add barriers
        os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 32)))
        _LIB = ctypes.CDLL(build())
        _LIB.orc_pool_out.restype = ctypes.c_int
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


def conv2d(x, w, b=None, stride=1, pad=0, dilate=1):
    x, xp = _f(x)
    w, wp = _f(w)
    N, C, H, W = x.shape
    K, Cw, kh, kw = w.shape
    assert Cw == C, (Cw, C)
    sh, sw = _pair(stride)
    ph, pw = _pair(pad)
    dh, dw = _pair(dilate)
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    y = np.empty((N, K, Ho, Wo), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    lib().orc_conv2d(xp, N, C, H, W, wp, bp, K, kh, kw, sh, sw, ph, pw, dh, dw,
                     y.ctypes.data_as(ctypes.c_void_p))
    return y


def deconv2d(x, w, b=None, stride=1, pad=0, groups=1):
    x, xp = _f(x)
    w, wp = _f(w)
    N, C, H, W = x.shape
    Cw, Kg, kh, kw = w.shape
    assert Cw == C
    K = Kg * groups
    sh, sw = _pair(stride)
    ph, pw = _pair(pad)
    Ho = sh * (H - 1) + kh - 2 * ph
    Wo = sw * (W - 1) + kw - 2 * pw
    y = np.empty((N, K, Ho, Wo), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    lib().orc_deconv2d(xp, N, C, H, W, wp, bp, K, groups, kh, kw, sh, sw, ph, pw,
                       y.ctypes.data_as(ctypes.c_void_p))
    return y


def deform_im2col(x, offset, kernel, stride, pad, dilate, dg):
    x, xp = _f(x)
    offset, op = _f(offset)
    C, H, W = x.shape[-3:]
    kh, kw = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(pad)
    dh, dw = _pair(dilate)
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    col = np.empty((C * kh * kw, Ho * Wo), np.float32)
    lib().orc_deform_im2col(xp, C, H, W, op, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                            col.ctypes.data_as(ctypes.c_void_p))
    return col


def deform_conv2d(x, offset, w, stride=1, pad=0, dilate=1, dg=1):
    x, xp = _f(x)
    offset, op = _f(offset)
    w, wp = _f(w)
    N, C, H, W = x.shape
    K, Cw, kh, kw = w.shape
    assert Cw == C
    sh, sw = _pair(stride)
    ph, pw = _pair(pad)
    dh, dw = _pair(dilate)
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    assert offset.shape == (N, 2 * kh * kw * dg, Ho, Wo), offset.shape
    y = np.empty((N, K, Ho, Wo), np.float32)
    ws = np.empty(C * kh * kw * Ho * Wo, np.float32)
    lib().orc_deform_conv2d(xp, N, C, H, W, op, wp, K, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                            ws.ctypes.data_as(ctypes.c_void_p),
                            y.ctypes.data_as(ctypes.c_void_p))
    return y


def deform_border_taps(H, W, offset, kernel, stride, pad, dilate, eps):
    """Output pixels (n, oy, ox) of a DeformableConvolution with at least one sampling position within `eps` of the
    DISCONTINUITY of the DCN-v1 rule (orc_deform_im2col above): a tap contributes 0 for h_im < 0 or h_im >= H and the
    border row's value just inside, likewise in w -- so a last-bit difference in an offset that sits within eps of 0 or
    of H (W) legitimately switches that tap on or off.  Returns an int array (k, 3).  float64 throughout."""
    off = np.asarray(offset, np.float64)
    N, ch, Ho, Wo = off.shape
    kh, kw = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(pad)
    dh, dw = _pair(dilate)
    dg = ch // (2 * kh * kw)
    off = off.reshape(N, dg, kh * kw, 2, Ho, Wo)
    i = (np.arange(kh * kw) // kw).reshape(1, 1, -1, 1, 1)
    j = (np.arange(kh * kw) % kw).reshape(1, 1, -1, 1, 1)
    oy = np.arange(Ho).reshape(1, 1, 1, -1, 1)
    ox = np.arange(Wo).reshape(1, 1, 1, 1, -1)
    h_im = oy * sh - ph + i * dh + off[:, :, :, 0]
    w_im = ox * sw - pw + j * dw + off[:, :, :, 1]
    near_h = (np.abs(h_im) < eps) | (np.abs(h_im - H) < eps)
    near_w = (np.abs(w_im) < eps) | (np.abs(w_im - W) < eps)
    live_h = (h_im > -eps) & (h_im < H + eps)
    live_w = (w_im > -eps) & (w_im < W + eps)
    crit = ((near_h & live_w) | (near_w & live_h)).any(axis=(1, 2))
    return np.argwhere(crit)


def bn_fold(gamma, beta, mean, var, eps, fix_gamma=False):
    gamma, gp = _f(gamma)
    beta, bp = _f(beta)
    mean, mp = _f(mean)
    var, vp = _f(var)
    C = beta.shape[0]
    scale = np.empty(C, np.float32)
    shift = np.empty(C, np.float32)
    lib().orc_bn_fold(gp, bp, mp, vp, C, ctypes.c_float(eps), int(bool(fix_gamma)),
                      scale.ctypes.data_as(ctypes.c_void_p),
                      shift.ctypes.data_as(ctypes.c_void_p))
    return scale, shift


def batchnorm(x, gamma, beta, mean, var, eps, fix_gamma=False):
    scale, shift = bn_fold(gamma, beta, mean, var, eps, fix_gamma)
    x, xp = _f(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_scale_shift(xp, N, C, H * W, scale.ctypes.data_as(ctypes.c_void_p),
                          shift.ctypes.data_as(ctypes.c_void_p),
                          y.ctypes.data_as(ctypes.c_void_p))
    return y


def pool2d(x, kind, kernel, stride, pad=0, convention="valid"):
    x, xp = _f(x)
    N, C, H, W = x.shape
    kh, kw = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(pad)
    full = int(convention == "full")
    Ho = lib().orc_pool_out(H, kh, sh, ph, full)
    Wo = lib().orc_pool_out(W, kw, sw, pw, full)
    y = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_pool2d(xp, N, C, H, W, int(kind == "max"), full, kh, kw, sh, sw, ph, pw,
                     y.ctypes.data_as(ctypes.c_void_p))
    return y


def grid_generator_warp(flow):
    flow, fp = _f(flow)
    N, two, H, W = flow.shape
    assert two == 2
    grid = np.empty_like(flow)
    lib().orc_grid_generator_warp(fp, N, H, W, grid.ctypes.data_as(ctypes.c_void_p))
    return grid


def bilinear_sampler(data, grid):
    data, dp = _f(data)
    grid, gp = _f(grid)
    N, C, H, W = data.shape
    Ho, Wo = grid.shape[2:]
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_bilinear_sampler(dp, N, C, H, W, gp, Ho, Wo,
                               out.ctypes.data_as(ctypes.c_void_p))
    return out


def flow_warp(feat, flow):
    """GridGenerator(warp) + BilinearSampler, accel_18.py:174-175."""
    return bilinear_sampler(feat, grid_generator_warp(flow))


def warp_border_points(flow):
    """Output pixels (n, y, x) of GridGenerator(warp) + BilinearSampler whose sampling position has a tap OUTSIDE the map while
    another is inside: x + flow_x in (-1, 0) or (W-1, W), likewise in y.  There the sampled value falls off linearly from the
    border pixel's value to the zero padding, so its derivative with respect to the sampling position is the FEATURE itself
    (hundreds) instead of a difference of neighbours -- and the reference's formula quantises that position: `flow + x` is
    rounded to the fp32 grid at x (7.6e-6 px at x = 127), `/ sx - 1` and `(g + 1) * (W-1) / 2` round again.  Two fp32
    evaluations whose flows differ in the last bits (any two summation orders) land on neighbouring grid values at a few
    per cent of such pixels, and the sampled value then differs by ulp(W-1) x |feature| (scripts/debug/flow_exact.py,
    profiles/r06_margin_bisect.log).  The parity report lists these pixels; it tolerates nothing extra there.
    Returns an int array (k, 3).  float64 throughout."""
    f = np.asarray(flow, np.float64)
    N, two, H, W = f.shape
    xr = f[:, 0] + np.arange(W).reshape(1, 1, W)
    yr = f[:, 1] + np.arange(H).reshape(1, H, 1)
    part_x = ((xr > -1) & (xr < 0)) | ((xr > W - 1) & (xr < W))
    part_y = ((yr > -1) & (yr < 0)) | ((yr > H - 1) & (yr < H))
    live_x = (xr > -1) & (xr < W)
    live_y = (yr > -1) & (yr < H)
    return np.argwhere((part_x & live_y) | (part_y & live_x))


def relu(x):
    return np.maximum(x, np.float32(0))


def leaky_relu(x, slope=0.1):
    x = np.asarray(x, np.float32)
    return np.where(x > 0, x, x * np.float32(slope)).astype(np.float32)


def crop_like(a, ref_hw, offset):
    """mx.symbol.Crop(*[a, b], offset=(oy, ox)): a[:, :, oy:oy+Hb, ox:ox+Wb]."""
    oy, ox = offset
    Hb, Wb = ref_hw
    out = a[:, :, oy:oy + Hb, ox:ox + Wb]
    assert out.shape[2:] == (Hb, Wb), (a.shape, ref_hw, offset)
    return np.ascontiguousarray(out)


def argmax_c(x):
    x, xp = _f(x)
    N, C, H, W = x.shape
    out = np.empty((N, H, W), np.uint8)
    lib().orc_argmax_c(xp, N, C, H * W, out.ctypes.data_as(ctypes.c_void_p))
    return out
