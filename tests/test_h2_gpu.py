"""The fp16x2 form of the fp32 layers (kernels.h ConvParams::wh2r / wubh; conv_b3r.hip NPL = 2, conv_wino_b3*.hip H2): every fp32
operand as two half terms, three fp16 MFMA products per multiply-add, operands centred in the half range by exact powers of two --
the weights per output channel on the host, the pixels by the scale a probed run measures (accel_plan_op_range).
Bars: against float64 the form must be as accurate as the bf16x3 form it replaces (both are at fp32 accumulation noise), whatever the
scale of the data; the calibration must put the largest pixel into [2^10, 2^11), keep its scale while the range is stable, move it
when the range moves, and report a range that outgrew the scale."""
import os

import numpy as np
import pytest

from accel_amd import runtime

pytestmark = pytest.mark.gpu

B3R = [76, 77, 79, 80, 81]
WINO = [41, 42, 43]


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def conv64(x, w, pad):
    """float64 3x3 / 1x1 stride-1 convolution of one image (NCHW), zero padding"""
    C, H, W = x.shape[1:]
    K, _, kh, kw = w.shape
    xp = np.zeros((C, H + 2 * pad, W + 2 * pad)); xp[:, pad:pad + H, pad:pad + W] = x[0]
    out = np.zeros((K, H, W))
    w64 = w.astype(np.float64)
    for ky in range(kh):
        for kx in range(kw):
            out += np.einsum('kc,chw->khw', w64[:, :, ky, kx], xp[:, ky:ky + H, kx:kx + W])
    return out[None]


class OneConv(object):
    """a one-convolution plan whose handle stays around (ranges, re-calibration, several runs)"""

    def __init__(self, ctx, cin, cout, H, W, k, tile, w):
        self.m = runtime.Model(ctx)
        self.shape = (cin, cout, H, W)
        al = lambda b: (b + 255) // 256 * 256
        o_y = al(H * W * cin * 4)
        self.m.set_param("w_weight", w)
        t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (o_y + al(H * W * cout * 4), cin * H * W * 4, cout * H * W * 4)
        t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d\n" % (cin, cin, H, W, cin, cin, H, W)
        t += "conv name=c in=A:0:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w_weight act=0 cin=%d cout=%d mode=conv tile=%d k=%d,%d s=1,1 p=%d,%d d=1,1\n" % (
            cin, cin, H, W, o_y, cout, cout, H, W, cin, cout, tile, k, k, k // 2, k // 2)
        t += "export_nchw src=A:%d:%d:%d:%d:%d dst=y:0:%d:%d:%d:%d\n" % (o_y, cout, cout, H, W, cout, cout, H, W)
        self.plan = self.m.add_plan("p", t)
        self.plan.finalize()

    def __call__(self, x):
        cin, cout, H, W = self.shape
        self.m.write("x", x)
        self.plan.run()
        return self.m.read("y", (1, cout, H, W))

    def close(self):
        self.m.close()


def run_form(ctx, form, monkeypatch, *args):
    monkeypatch.setenv("ACCEL_SPLIT", form)
    c = OneConv(ctx, *args)
    return c


@pytest.mark.parametrize("tile", B3R + WINO)
def test_fp16x2_is_as_accurate_as_bf16x3_at_every_scale(ctx, tile, monkeypatch):
    k = 3 if tile in WINO else 1
    cin, cout, H, W = (128, 136, 24, 32) if tile in WINO else (1024, 136, 24, 32)
    for xscale, wscale in ((1.0, 0.03), (1e-4, 2.0), (3e3, 1e-3), (1e-7, 1e-3)):
        rng = np.random.default_rng(7)
        x = (np.maximum(rng.standard_normal((1, cin, H, W)), 0) * xscale * np.exp(rng.standard_normal((1, cin, 1, 1)))).astype(np.float32)
        w = (rng.standard_normal((cout, cin, k, k)) * wscale * np.exp(rng.standard_normal((cout, 1, 1, 1)))).astype(np.float32)
        ref = conv64(x, w, k // 2)
        s = np.abs(ref).max()
        err = {}
        for form in ("b3", "h2"):
            c = run_form(ctx, form, monkeypatch, cin, cout, H, W, k, tile, w)
            try:
                assert [o["mode"] for o in c.plan.ops() if o["kind"] == "conv"] == [3 if form == "h2" else 0]
                err[form] = float(np.abs(c(x) - ref).max() / s)
            finally:
                c.close()
        # both at fp32 accumulation noise (Winograd: about 3x a direct evaluation); the new form within 1.5x of the old one
        assert err["h2"] <= (3e-6 if tile in WINO else 1e-6), (tile, xscale, err)
        assert err["h2"] <= 1.5 * err["b3"] + 1e-7, (tile, xscale, err)


def test_b3r_geometries_agree_bit_for_bit_in_the_fp16x2_form(ctx, monkeypatch):
    """same split, same products in the same order, same K order: only the tiling differs (M large enough that nothing splits K)"""
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    x, w = np.maximum(rnd(1, 1, 64, 160, 256), 0), rnd(2, 128, 64, 1, 1, scale=0.1)
    outs = []
    for tile in B3R:
        c = OneConv(ctx, 64, 128, 160, 256, 1, tile, w)
        try:
            outs.append(c(x))
        finally:
            c.close()
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_calibration_scale_hysteresis_and_report(ctx, monkeypatch):
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    monkeypatch.setenv("ACCEL_RECAL_EVERY", "0")          # only the first run and explicit re-calibrations probe
    cin, cout, H, W = 256, 128, 16, 32
    w = rnd(3, cout, cin, 1, 1, scale=0.05)
    x = np.maximum(rnd(4, 1, cin, H, W), 0)
    c = OneConv(ctx, cin, cout, H, W, 1, 81, w)
    try:
        assert c.plan.ranges() == {"c": (1.0, False)}                      # before the first run
        ref = conv64(x, w, 0)
        tol = 1e-6 * np.abs(ref).max()
        assert np.abs(c(x) - ref).max() <= tol
        s0, cal = c.plan.ranges()["c"]
        amax = float(x.max())
        assert cal and 2.0 ** 10 <= s0 * amax < 2.0 ** 11 and np.log2(s0) == int(np.log2(s0))
        # a range that moved by 1.5x keeps the scale even when probed (inside the hysteresis window [2^8, 2^12))
        c.plan.recalibrate()
        assert np.abs(c(1.5 * x) - 1.5 * ref).max() <= 1.5 * tol and c.plan.ranges()["c"][0] == s0
        # 1/1000 of the range, not re-calibrated: still computed (lo terms lose bits -- the reason the probe exists) ...
        y = c(1e-3 * x)
        assert np.abs(y - 1e-3 * ref).max() <= 2e-3 * np.abs(ref).max() * 1e-3
        # ... and with a probe the scale follows and the result is at full accuracy again
        c.plan.recalibrate()
        assert np.abs(c(1e-3 * x) - 1e-3 * ref).max() <= 1e-3 * tol
        s1 = c.plan.ranges()["c"][0]
        assert 2.0 ** 10 <= s1 * amax * 1e-3 < 2.0 ** 11
        # the range outgrows the scale in force (x 1e6 since the last probe): the probe corrects the scale for THIS frame and the
        # library reports that the frames since the last probe may have saturated -- once
        c.plan.recalibrate()
        y = c(1e3 * x)
        assert np.abs(y - 1e3 * ref).max() <= 1e3 * tol
        with pytest.raises(runtime.AccelError, match="outgrew the half range"):
            c(1e3 * x)
        assert np.abs(c(1e3 * x) - 1e3 * ref).max() <= 1e3 * tol              # reported once; the plan keeps working
        # a non-finite input is reported as well
        bad = x.copy(); bad[0, 3, 2, 1] = np.nan
        c.plan.recalibrate()
        c(bad)
        with pytest.raises(runtime.AccelError, match="not finite"):
            c(x)
    finally:
        c.close()


def test_periodic_recalibration(ctx, monkeypatch):
    """ACCEL_RECAL_EVERY = 3: runs 0, 3, 6 ... of a plan are probed"""
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    monkeypatch.setenv("ACCEL_RECAL_EVERY", "3")
    cin, cout, H, W = 64, 64, 16, 16
    w, x = rnd(5, cout, cin, 1, 1, scale=0.1), np.maximum(rnd(6, 1, cin, H, W), 0)
    c = OneConv(ctx, cin, cout, H, W, 1, 76, w)
    try:
        c(x)                                   # run 0: probed
        s0 = c.plan.ranges()["c"][0]
        c(x * 2.0 ** -10); c(x * 2.0 ** -10)    # runs 1, 2: not probed
        assert c.plan.ranges()["c"][0] == s0
        c(x * 2.0 ** -10)                       # run 3: probed, the range moved by 2^-10
        assert c.plan.ranges()["c"][0] == s0 * 2.0 ** 10
    finally:
        c.close()


def test_split_b3_plan_option_and_env(ctx, monkeypatch):
    """ACCEL_SPLIT=b3 (or plan option split=b3) keeps the range-free bf16x3 form: no scale slots, mode 0"""
    monkeypatch.setenv("ACCEL_SPLIT", "b3")
    w = rnd(8, 64, 64, 1, 1, scale=0.1)
    c = OneConv(ctx, 64, 64, 16, 16, 1, 76, w)
    try:
        assert c.plan.ranges() == {} and [o["mode"] for o in c.plan.ops() if o["kind"] == "conv"] == [0]
    finally:
        c.close()


def test_stem_fp16x2_is_as_accurate_as_bf16x3(ctx, monkeypatch):
    """the 7x7/2 stem (geometry 51, conv_stem_b3.hip) in both forms against a float64 convolution: images of the scale the
    reference feeds (mean-subtracted 8-bit pixels), tiny ones and large ones; ragged tiles"""
    import torch
    import torch.nn.functional as F
    for xscale, wscale in ((50.0, 0.002), (1e-3, 1.0), (2e3, 1e-4)):
        x, w = rnd(90, 2, 3, 120, 200, scale=xscale), rnd(91, 64, 3, 7, 7, scale=wscale)
        truth = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride=2, padding=3).numpy()
        sc = float(np.abs(truth).max())
        err = {}
        for form in ("b3", "h2"):
            monkeypatch.setenv("ACCEL_SPLIT", form)
            err[form] = float(np.abs(ctx.conv2d(x, w, None, 2, 3, 1, tile=51) - truth).max()) / sc
        print("stem vs float64 at pixel scale %g: bf16x3 %.2e, fp16x2 %.2e" % (xscale, err["b3"], err["h2"]))
        assert err["h2"] <= 3e-6 and err["h2"] <= 1.5 * err["b3"] + 1e-7, (xscale, err)


def test_a_layer_left_without_a_range_is_probed_again(ctx, monkeypatch):
    """an all-zero input says nothing about the range: the probed first run leaves the layer uncalibrated (scale 1), and the runs that
    follow are probed as well until it has one -- it does not wait for the periodic re-calibration (here: never)"""
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    monkeypatch.setenv("ACCEL_RECAL_EVERY", "0")
    cin, cout, H, W = 256, 128, 16, 32
    w = rnd(3, cout, cin, 1, 1, scale=0.05)
    x = np.maximum(rnd(4, 1, cin, H, W), 0) * np.float32(1e-4)          # small pixels: at scale 1 the lo terms would be sub-normal halves
    c = OneConv(ctx, cin, cout, H, W, 1, 81, w)
    try:
        assert np.array_equal(c(np.zeros_like(x)), np.zeros((1, cout, H, W), np.float32))
        assert c.plan.ranges()["c"] == (1.0, False)
        ref = conv64(x, w, 0)
        y = c(x)                                                            # probed again: the scale follows THIS frame
        s, cal = c.plan.ranges()["c"]
        assert cal and 2.0 ** 10 <= s * float(x.max()) < 2.0 ** 11
        assert np.abs(y - ref).max() <= 1e-6 * np.abs(ref).max()
        c(x)
        assert c.plan.ranges()["c"] == (s, True)                            # and the runs after that are ordinary ones
    finally:
        c.close()
