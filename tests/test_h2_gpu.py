"""The fp16x2 form of the fp32 layers (kernels.h ConvParams::wh2r / wubh; conv_b3r.hip NPL = 2, conv_wino_b3*.hip H2): every fp32
operand as two half terms, three fp16 MFMA products per multiply-add, operands centred in the half range by exact powers of two --
the weights per output channel on the host, the pixels by the scale the convolution derives in its prologue from the range slot of its
input tensor, raised IN THE SAME RUN by whoever wrote that tensor (csrc/range.h; accel_plan_op_range).
Bars: against float64 the form must be as accurate as the bf16x3 form it replaces (both are at fp32 accumulation noise), whatever the
scale of the data; the scale must put the largest pixel of THIS run into [2^13, 2^14) whatever the runs before it looked like (a plan
run is a pure function of its inputs: no calibration, no history), and a non-finite input must be reported."""
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


def test_the_scale_follows_every_run_and_no_run_remembers_another(ctx, monkeypatch):
    """One bound plan, inputs whose range jumps by 1e-3 ... 1e6 between consecutive runs: every run at full accuracy, the scale a
    function of that run's input alone, the same input bit-identical whatever came before it; non-finite inputs reported."""
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    cin, cout, H, W = 256, 128, 16, 32
    w = rnd(3, cout, cin, 1, 1, scale=0.05)
    x = np.maximum(rnd(4, 1, cin, H, W), 0)
    c = OneConv(ctx, cin, cout, H, W, 1, 81, w)
    try:
        assert c.plan.ranges() == {"c": (1.0, 0)}                          # before the first run
        ref = conv64(x, w, 0)
        tol = 1e-6 * np.abs(ref).max()
        first = c(x)
        assert np.abs(first - ref).max() <= tol
        s0, src = c.plan.ranges()["c"]
        amax = float(x.max())
        assert src == 2 and 2.0 ** 13 <= s0 * amax < 2.0 ** 14 and np.log2(s0) == int(np.log2(s0))      # (import_nchw has no epilogue: measured)
        for f in (1.5, 1e-3, 1e3, 1e-6, 1e6 * 1e-6, 100.0, 1.0):          # consecutive runs, no re-binding, ranges jumping both ways
            y = c(np.float32(f) * x)
            s, _ = c.plan.ranges()["c"]
            assert 2.0 ** 13 <= s * amax * np.float32(f) < 2.0 ** 14, (f, s)
            assert np.abs(y - f * ref).max() <= f * tol * 1.01, f
        assert np.array_equal(c(x), first)                                  # the same frame after another history: bit-identical
        assert np.array_equal(c(np.zeros_like(x)), np.zeros((1, cout, H, W), np.float32))
        assert c.plan.ranges()["c"] == (1.0, 0)                             # an all-zero input: scale 1
        assert np.array_equal(c(x), first)
        # a non-finite input is reported by the next run (once); the plan keeps working
        bad = x.copy(); bad[0, 3, 2, 1] = np.inf
        c(bad)
        with pytest.raises(runtime.AccelError, match="not finite"):
            c(x)
        assert np.array_equal(c(x), first)
    finally:
        c.close()


def _two_convs(ctx, cin, cmid, cout, H, W, w1, w2, tile2, graph=False):
    """conv 1x1 (fp32 geometry 0, ReLU) -> conv 1x1 (forced fp16x2 geometry): the second one's range comes from the first one's epilogue"""
    m = runtime.Model(ctx)
    al = lambda b: (b + 255) // 256 * 256
    o_m, o_y = al(H * W * cin * 4), al(H * W * cin * 4) + al(H * W * cmid * 4)
    m.set_param("w1_weight", w1); m.set_param("w2_weight", w2)
    t = ("" if graph else "option graph=0\n") + "option tune=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (
        o_y + al(H * W * cout * 4), cin * H * W * 4, cout * H * W * 4)
    t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d\n" % (cin, cin, H, W, cin, cin, H, W)
    t += "conv xr=0 yr=1 name=a in=A:0:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w1_weight act=1 cin=%d cout=%d mode=conv tile=0 k=1,1 s=1,1 p=0,0 d=1,1\n" % (
        cin, cin, H, W, o_m, cmid, cmid, H, W, cin, cmid)
    t += "conv xr=1 yr=2 name=b in=A:%d:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w2_weight act=0 cin=%d cout=%d mode=conv tile=%d k=1,1 s=1,1 p=0,0 d=1,1\n" % (
        o_m, cmid, cmid, H, W, o_y, cout, cout, H, W, cmid, cout, tile2)
    t += "export_nchw src=A:%d:%d:%d:%d:%d dst=y:0:%d:%d:%d:%d\n" % (o_y, cout, cout, H, W, cout, cout, H, W)
    plan = m.add_plan("p", t)
    plan.finalize()
    return m, plan


@pytest.mark.parametrize("graph", [False, True])
def test_the_producer_epilogue_raises_the_consumers_range(ctx, monkeypatch, graph):
    """a -> b inside one plan (eager and as a captured graph): b's scale comes from the range a's epilogue noted (source 1, no pass over
    the tensor), equals the scale of the tensor's true maximum, follows every run, and b is at full accuracy"""
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    cin, cmid, cout, H, W = 64, 256, 128, 24, 40
    w1, w2 = rnd(11, cmid, cin, 1, 1, scale=0.2), rnd(12, cout, cmid, 1, 1, scale=0.05)
    x = rnd(13, 1, cin, H, W)
    m, plan = _two_convs(ctx, cin, cmid, cout, H, W, w1, w2, 81, graph=graph)
    try:
        outs = {}
        for f in (1.0, 1e-4, 300.0, 1.0):
            m.write("x", np.float32(f) * x)
            plan.run()
            y = m.read("y", (1, cout, H, W))
            mid = np.maximum(conv64(np.float32(f) * x, w1, 0), 0)
            ref = conv64(mid.astype(np.float32), w2, 0)
            s, src = plan.ranges()["b"]
            assert src == 1 and plan.ranges()["a"] == (1.0, 0)         # b: from a's epilogue; a runs on the fp32 MFMA (geometry 0): it reads no range
            assert 2.0 ** 13 <= s * float(mid.max()) * (1 + 1e-6) and s * float(mid.max()) * (1 - 1e-6) < 2.0 ** 14, (f, s, mid.max())
            assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max(), f
            outs.setdefault(f, y)
            assert np.array_equal(y, outs[f])                           # f = 1.0 twice, with other frames between: bit-identical
    finally:
        m.close()


def test_what_is_reported_when_a_producer_emits_non_finite_values(ctx, monkeypatch):
    """include/accel_hip.h (accel_plan_op_range), round-5 advisor: an INFINITY a matrix-core convolution writes mid-plan is seen by its
    consumer's range slot and reported by the next run; a NaN is NOT -- the epilogue's floating-point maxima drop NaNs: behind a ReLU the
    NaN becomes 0 (fmaxf; the reference's relu `a > 0 ? a : 0` does the same), without one it reaches the outputs through the matrix
    instructions as in fp32 arithmetic -- and no error is raised.  (A NaN in a tensor that is MEASURED or written by a byte mover is
    reported: test_the_scale_follows_every_run... covers the measured case.)"""
    monkeypatch.setenv("ACCEL_SPLIT", "h2")
    cin, cmid, cout, H, W = 64, 256, 128, 24, 40
    w1, w2 = rnd(11, cmid, cin, 1, 1, scale=0.2), rnd(12, cout, cmid, 1, 1, scale=0.05)
    x = rnd(13, 1, cin, H, W)
    m, plan = _two_convs(ctx, cin, cmid, cout, H, W, w1, w2, 81)
    try:
        m.write("x", x); plan.run()
        good = m.read("y", (1, cout, H, W)).copy()
        bad = x.copy(); bad[0, 5, 3, 7] = np.nan             # conv a (fp32 MFMA, ReLU): every channel of that pixel is relu(NaN) = 0
        m.write("x", bad); plan.run()
        y = m.read("y", (1, cout, H, W)).copy()
        assert np.isfinite(y).all() and not y[0, :, 3, 7].any()
        y[0, :, 3, 7] = good[0, :, 3, 7]
        assert np.array_equal(y, good)                          # ... and nothing else moved (same range, same scale)
        m.write("x", x); plan.run()                             # no ACCEL_ERR_RANGE: the NaN never entered b's range slot
        assert np.array_equal(m.read("y", (1, cout, H, W)), good)
        big = x.copy(); big[0, 5, 3, 7] = np.inf               # a's output is +inf / 0 (ReLU) at that pixel: b's slot holds an infinity
        m.write("x", big); plan.run()
        m.write("x", x)
        with pytest.raises(runtime.AccelError, match="an infinity"):
            plan.run()
        plan.run()
        assert np.array_equal(m.read("y", (1, cout, H, W)), good)
    finally:
        m.close()


def test_split_b3_plan_option_and_env(ctx, monkeypatch):
    """ACCEL_SPLIT=b3 (or plan option split=b3) keeps the bf16x3 form: no range slots, mode 0"""
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
