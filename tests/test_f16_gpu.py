"""fp16-MFMA convolution mode (BASELINE config 5 "fp16 convs"): fp32 activations in HBM,
operands rounded to half in the loader, fp32 accumulation.  Tolerance is the half-precision
operand rounding (2^-11 relative per operand), not the fp32 bar."""
import os

import numpy as np
import pytest

from accel_amd.utils import image, synth
from oracle import graphs as G, ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def f16_mode():
    os.environ["ACCEL_CONV_DTYPE"] = "f16"
    yield
    os.environ.pop("ACCEL_CONV_DTYPE", None)


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def h(a):
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 10, 76, 77, 79, 80, 81])
@pytest.mark.parametrize("C,K,H,W,k,s,p,d", [(64, 136, 23, 31, 3, 2, 2, 2), (256, 72, 9, 13, 1, 1, 0, 1), (128, 200, 20, 36, 3, 1, 1, 1),
                                            (40, 300, 17, 19, 3, 1, 1, 1)])
def test_conv_f16_matches_half_rounded_operands(ctx, f16_mode, tile, C, K, H, W, k, s, p, d):
    x, w, b = rnd(1, 1, C, H, W), rnd(2, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(3, K)
    got = ctx.conv2d(x, w, b, s, p, d, tile=tile)
    ref = O.conv2d(h(x), h(w), b, s, p, d)          # exact products of the rounded operands, fp32 sums
    # measured 5e-7 ... 2.5e-6 of the output range (scripts/debug/f16_probe.py): the mode IS "operands rounded to half, fp32
    # products and sums" -- only the fp32 summation order differs from the oracle's evaluation of that specification
    assert float(np.abs(got - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    full = O.conv2d(x, w, b, s, p, d)
    assert float(np.abs(got - full).max()) <= 5e-3 * max(1.0, float(np.abs(full).max()))


@pytest.mark.parametrize("tile", [0, 76, 77, 79, 80, 81])
def test_f16_register_weight_kernel_deconv_splitk_and_batch(ctx, f16_mode, tile):
    """The one-plane fp16 form of conv_b3r.hip (76 / 77 / 79 / 80 / 81 on an f16 layer: half-rounded weights in MFMA fragment
    order global -> VGPR, pixels rounded by the loader into ONE LDS plane) on the four parity classes of the 4x4/2
    deconvolution, split-K, a batch of images and an odd number of K steps -- to the same 1e-5 of "rounded operands, fp32
    sums" as the fp16 kernel of conv_igemm.hip (tile 0, the control)."""
    from accel_amd import runtime
    cin, cout, H, W = 96, 160, 9, 13
    x, w = rnd(40, cin, H, W), rnd(41, cin, cout, 4, 4, scale=0.05)
    m = runtime.Model(ctx)
    try:
        m.set_param("w_weight", w)
        al = lambda b: (b + 255) // 256 * 256
        o_y = al(H * W * cin * 4)
        t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (o_y + al(4 * H * W * cout * 4), cin * H * W * 4, cout * 4 * H * W * 4)
        t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d\n" % (cin, cin, H, W, cin, cin, H, W)
        t += "conv name=c in=A:0:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w_weight act=0 cin=%d cout=%d mode=deconv2x tile=%d\n" % (
            cin, cin, H, W, o_y, cout, cout, 2 * H, 2 * W, cin, cout, tile)
        t += "export_nchw src=A:%d:%d:%d:%d:%d dst=y:0:%d:%d:%d:%d\n" % (o_y, cout, cout, 2 * H, 2 * W, cout, cout, 2 * H, 2 * W)
        plan = m.add_plan("p", t)
        m.write("x", x)
        plan.finalize()
        plan.run()
        got = m.read("y", (1, cout, 2 * H, 2 * W))
    finally:
        m.close()
    ref = O.deconv2d(h(x[None]), h(w), None, 2, 1)
    assert float(np.abs(got - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    for (N, C, K, Hh, Ww, k, s_, p_, d_) in ((1, 512, 136, 8, 16, 3, 1, 1, 1), (3, 64, 72, 20, 28, 3, 1, 2, 2), (2, 96, 64, 12, 10, 5, 2, 2, 1)):
        xx, ww, bb = rnd(42, N, C, Hh, Ww), rnd(43, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(44, K)
        ref = O.conv2d(h(xx), h(ww), bb, s_, p_, d_)
        got = ctx.conv2d(xx, ww, bb, s_, p_, d_, tile=tile)
        assert float(np.abs(got - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max())), (N, C, K)


def test_deconv_and_dcn_f16(ctx, f16_mode):
    x, w = rnd(4, 1, 64, 9, 13), rnd(5, 64, 32, 4, 4, scale=0.1)
    ref = O.deconv2d(h(x), h(w), None, 2, 1)
    assert float(np.abs(ctx.deconv2d_4x4s2(x, w) - ref).max()) <= 2e-4 * max(1.0, float(np.abs(ref).max()))
    xd, wd, off = rnd(6, 1, 32, 12, 20), rnd(7, 48, 32, 3, 3, scale=0.06), rnd(8, 1, 72, 12, 20)
    ref = O.deform_conv2d(xd, off, wd, 1, 2, 2, 4)
    assert float(np.abs(ctx.deform_conv2d(xd, off, wd, 1, 2, 2, 4) - ref).max()) <= 5e-3 * max(1.0, float(np.abs(ref).max()))


def test_clip_f16_close_to_fp32_oracle(demo_cfg, f16_mode):
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 128, 256, 3
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("50", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 3)
    try:
        outs = demo.run_clip("50", demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, "50", [image.transform(f, demo_cfg.network.PIXEL_MEANS).astype(np.float32) for f in frames], interval)
    for t, ((lg, lab), (rlg, rlab)) in enumerate(zip(outs, ref)):
        # ~100 layers of half-rounded operands on random weights: measured max error 4-5 % of the logit
        # range at the worst pixel, mean error 20x smaller, 0.1-0.2 % of the labels differ
        scale = max(1.0, float(np.abs(rlg).max()))
        assert float(np.abs(lg - rlg).max()) <= 0.1 * scale, "frame %d" % t
        assert float(np.abs(lg - rlg).mean()) <= 1e-2 * scale, "frame %d" % t
        assert float((lab != rlab[0]).mean()) < 1e-2


@pytest.mark.gpu_extra      # (the f16 mode WITHOUT half storage, a non-default sub-mode; the default is checked at this size in test_f16_storage_gpu.py and at config 5's)
def test_clip_f16_matches_the_oracle_on_half_rounded_operands_512x1024(demo_cfg, f16_mode, monkeypatch):
    """The reduced-precision mode against ITS OWN specification: oracle.graphs with ROUND_F16 rounds the operands of the
    same layers the HIP loader rounds (Cin % 8 == 0, more than 4 output channels; deformable layers: the sampled columns)
    and keeps products and sums in fp32.  Accel-50 (BASELINE config 5's model), 512x1024, key + non-key frame, the
    reference's layer list one to one.

    What this can and cannot show.  PER LAYER the mode meets that specification to fp32 rounding (1e-5 bar in
    test_conv_f16_matches_half_rounded_operands, measured 5e-7 ... 2.5e-6).  Over ~100 layers two evaluations of the same
    specification nevertheless drift apart: rounding to half is discontinuous, an operand that sits within fp32 noise of a
    rounding boundary goes to the other neighbour in one of the two evaluations, that difference (half an fp16 ulp) makes
    more operands of the next layer flip, and the trajectories decorrelate until their distance is of the order of the
    half-precision noise itself.  Measured (512x1024, key frame; two tune tables): median |error| 7.6e-5 of the logit range
    against the rounded-operand oracle, 1.0e-4 against the fp32 oracle; the 99 % / 99.9 % quantiles and the maximum (about
    5e-4 / 6e-3 / 4e-2: the deformable layers' border discontinuity reacting to offsets that carry half-precision noise)
    are the same in both comparisons within run-to-run differences.  So the whole-graph assertions are only: the typical
    pixel is no farther from the mode's own specification than from the fp32 result, the worst pixel stays inside 10 % of
    the logit range, fewer than 0.5 % of the labels differ.  The quantiles are printed for the log."""
    from accel_amd import demo
    from accel_amd.core import tester
    monkeypatch.setenv("ACCEL_FOLD_LINEAR", "0")
    H, W, interval = 512, 1024, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("50", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip("50", demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    fr = [image.transform(f, demo_cfg.network.PIXEL_MEANS).astype(np.float32) for f in frames]
    G.ROUND_F16 = True
    try:
        ref16 = G.run_clip(P, "50", fr, interval)
    finally:
        G.ROUND_F16 = False
    ref32 = G.run_clip(P, "50", fr, interval)
    for t, ((lg, lab), (r16, l16), (r32, l32)) in enumerate(zip(outs, ref16, ref32)):
        scale = max(1.0, float(np.abs(r32).max()))
        d16, d32 = np.abs(lg - r16).ravel() / scale, np.abs(lg - r32).ravel() / scale
        q = lambda d: tuple(float(np.quantile(d[::7], p_)) for p_ in (0.5, 0.99, 0.999)) + (float(d.max()),)
        q16, q32 = q(d16), q(d32)
        print("f16 mode frame %d, |error| / logit range (median, 99 %%, 99.9 %%, max): vs the oracle on half-rounded operands %.2e %.2e %.2e %.2e | "
              "vs the fp32 oracle %.2e %.2e %.2e %.2e; labels differing %.4f %% / %.4f %%"
              % ((t,) + q16 + q32 + (100 * float((lab != l16[0]).mean()), 100 * float((lab != l32[0]).mean()))))
        assert q16[0] <= 1.05 * q32[0], "frame %d" % t
        assert q16[3] <= 0.1 and float((lab != l16[0]).mean()) < 5e-3, "frame %d" % t
