"""fp32 convolutions on the bf16 matrix cores ("bf16x3", ACCEL_CONV_DTYPE=bf16x3): every fp32 operand is split exactly
into three bf16 terms and a product is taken as six bf16 x bf16 products accumulated in fp32 (conv_igemm.hip).  The claim
tested here is "fp32-equivalent": against a float64 evaluation of the same convolution the error is of the order of
the fp32-MFMA kernel's own (both are dominated by fp32 accumulation), far inside the 1e-4 operator tolerance and the
1e-3 logit tolerance of the parity bar -- unlike the fp16 mode, which misses them by two orders of magnitude."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from accel_amd.utils import synth
from oracle import graphs as G, ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def b3_mode():
    os.environ["ACCEL_CONV_DTYPE"] = "bf16x3"
    yield
    os.environ.pop("ACCEL_CONV_DTYPE", None)


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def conv64(x, w, b, s, p, d):
    y = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None if b is None else torch.from_numpy(b).double(),
                 stride=s, padding=p, dilation=d)
    return y.numpy()


CASES = [(64, 136, 23, 31, 3, 2, 2, 2), (256, 72, 9, 13, 1, 1, 0, 1), (128, 200, 20, 36, 3, 1, 1, 1),
         (512, 256, 16, 24, 3, 1, 1, 1),      # K = 4608: the longest reductions of the path
         (1024, 19 * 4, 8, 16, 1, 1, 0, 1)]


@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 10])
@pytest.mark.parametrize("C,K,H,W,k,s,p,d", CASES[:3])
def test_conv_bf16x3_every_geometry_within_the_fp32_tolerance(ctx, b3_mode, tile, C, K, H, W, k, s, p, d):
    x, w, b = rnd(1, 2, C, H, W), rnd(2, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(3, K)
    got = ctx.conv2d(x, w, b, s, p, d, tile=tile)
    ref = O.conv2d(x, w, b, s, p, d)
    assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))     # the fp32 operator bar of test_ops_gpu


@pytest.mark.parametrize("tile", [70, 71, 72, 73, 74, 75, 76, 77, 79, 80, 81])
def test_bf16x3_geometries_of_an_fp32_layer(ctx, tile):
    """In fp32 mode the kernel is offered to the autotuner as launch geometries 70-74 of the same convolution."""
    from accel_amd.runtime import AccelError
    x, w, b, res = rnd(20, 2, 64, 23, 31), rnd(21, 136, 64, 3, 3, scale=0.05), rnd(22, 136), rnd(23, 2, 136, 12, 16)
    ref = O.relu(O.conv2d(x, w, b, 2, 2, 2) + res)
    got = ctx.conv2d(x, w, b, 2, 2, 2, residual=res, act=1, tile=tile)
    assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    # channel counts that are not multiples of 8: the two 4-wide granules of a lane's chunk may lie in different taps
    for C in (36, 3, 20):
        xc, wc = rnd(28, 1, C, 17, 19), rnd(29, 40, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5)
        refc = O.conv2d(xc, wc, None, 1, 1, 1)
        assert float(np.abs(ctx.conv2d(xc, wc, None, 1, 1, 1, tile=tile) - refc).max()) <= 1e-4 * max(1.0, float(np.abs(refc).max())), C
    with pytest.raises(AccelError, match="bf16x3"):
        ctx.conv2d(rnd(26, 1, 16, 16, 16), rnd(27, 2, 16, 3, 3), None, 1, 1, 1, tile=tile)      # 2 output channels: the strip kernel's layer


@pytest.mark.parametrize("tile", [74, 76, 77, 79, 80, 81])
def test_bf16x3_register_weight_kernel_deconv_splitk_and_batch(ctx, tile):
    """conv_b3r.hip (76, 79, 80, 81: weight fragments global -> VGPR in MFMA order, pixel tile double-buffered in LDS)
    on what the convolution test above does not reach: the four parity classes of the 4x4/2 deconvolution (class-major
    planes), split-K (small pixel count, long reduction: partial sums + reduce kernel), a batch of images in one GEMM,
    and an odd number of K steps.  74 (the first-generation kernel) runs the same cases as the control."""
    from accel_amd import runtime
    # deconvolution 4x4/2 through a one-op plan with the geometry forced
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
    ref = O.deconv2d(x[None], w, None, 2, 1)
    assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    # split-K: 8 x 16 pixels, K = 9 * 512; batch of 3 images; K = 5 * 96 = 15 K steps (odd)
    for (N, C, K, Hh, Ww, k, s_, p_, d_) in ((1, 512, 136, 8, 16, 3, 1, 1, 1), (3, 64, 72, 20, 28, 3, 1, 2, 2), (2, 96, 64, 12, 10, 5, 2, 2, 1)):
        xx, ww, bb = rnd(42, N, C, Hh, Ww), rnd(43, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(44, K)
        ref = O.conv2d(xx, ww, bb, s_, p_, d_)
        got = ctx.conv2d(xx, ww, bb, s_, p_, d_, tile=tile)
        assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max())), (N, C, K)


@pytest.mark.parametrize("C,K,H,W,k,s,p,d", CASES)
def test_conv_bf16x3_error_is_of_the_order_of_the_fp32_kernels(ctx, C, K, H, W, k, s, p, d):
    x, w, b = rnd(4, 1, C, H, W, scale=3.0), rnd(5, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(6, K)
    truth = conv64(x, w, b, s, p, d)
    scale = max(1.0, float(np.abs(truth).max()))
    e32 = float(np.abs(ctx.conv2d(x, w, b, s, p, d) - truth).max()) / scale
    os.environ["ACCEL_CONV_DTYPE"] = "bf16x3"
    try:
        eb3 = float(np.abs(ctx.conv2d(x, w, b, s, p, d) - truth).max()) / scale
    finally:
        os.environ.pop("ACCEL_CONV_DTYPE", None)
    os.environ["ACCEL_CONV_DTYPE"] = "f16"
    try:
        e16 = float(np.abs(ctx.conv2d(x, w, b, s, p, d) - truth).max()) / scale
    finally:
        os.environ.pop("ACCEL_CONV_DTYPE", None)
    print("K=%d: max error / output scale vs float64: fp32 MFMA %.2e, bf16x3 %.2e, fp16 %.2e" % (C * k * k, e32, eb3, e16))
    assert eb3 <= 3e-6                       # fp32 rounding level (2^-23 = 1.2e-7 per operand, sqrt(K)-ish growth)
    assert eb3 <= 4 * e32 + 2e-7             # same order as the fp32 kernel
    assert e16 >= 20 * eb3                   # and not the fp16 mode's


def test_deconv_and_dcn_bf16x3(ctx, b3_mode):
    xo, wo = rnd(14, 2, 386, 8, 12), rnd(15, 386, 64, 4, 4, scale=0.03)      # FlowNet deconv2: 386 = 256 + 128 + 2 channels
    refo = O.deconv2d(xo, wo, None, 2, 1)
    assert float(np.abs(ctx.deconv2d_4x4s2(xo, wo) - refo).max()) <= 1e-4 * max(1.0, float(np.abs(refo).max()))
    x, w = rnd(4, 1, 64, 9, 13), rnd(5, 64, 32, 4, 4, scale=0.1)
    ref = O.deconv2d(x, w, None, 2, 1)
    assert float(np.abs(ctx.deconv2d_4x4s2(x, w) - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    xd, wd, off = rnd(6, 1, 32, 12, 20), rnd(7, 48, 32, 3, 3, scale=0.06), rnd(8, 1, 72, 12, 20)
    ref = O.deform_conv2d(xd, off, wd, 1, 2, 2, 4)
    assert float(np.abs(ctx.deform_conv2d(xd, off, wd, 1, 2, 2, 4) - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("version,interval", [("18", 2), ("101", 2)])
def test_clip_bf16x3_meets_the_fp32_parity_bar(demo_cfg, b3_mode, version, interval):
    """The whole-graph check of test_graph_gpu, unchanged: logits within 1e-3 of the fp32 oracle, labels identical
    wherever the oracle's top-2 margin exceeds twice the measured error."""
    from accel_amd import demo
    from accel_amd.core import tester
    from accel_amd.utils import image
    from parity_report import check_against_oracle
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    frames = synth.make_clip(H, W, 3)
    try:
        outs = demo.run_clip(version, demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, version, [image.transform(f, demo_cfg.network.PIXEL_MEANS).astype(np.float32) for f in frames], interval)
    check_against_oracle(outs, ref, "bf16x3 accel-" + version)
