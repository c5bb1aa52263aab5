"""conv_b3d.hip (launch geometries 82-89): the implicit GEMM of an fp32 layer on the 16-bit matrix cores with both operands brought
in by LDS-DMA and the pixel operand split after its fragment read -- in the bf16x3 form (82-87) and in the fp16x2 form (82-89).
Same arithmetic as conv_b3r.hip in the same form (same split, the products in the same order, the K steps in the same order), so
besides the operator bar against the oracle the results must be BIT-IDENTICAL to geometry 81 wherever neither launch splits K."""
import os

import numpy as np
import pytest

from oracle import ops as O

pytestmark = pytest.mark.gpu

B3D = [("b3", t) for t in (82, 83, 84, 85, 86, 87)] + [("h2", t) for t in (82, 83, 84, 85, 86, 87, 88, 89)]
form_tile = pytest.mark.parametrize("form,tile", B3D)


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def close(got, ref, what=None):
    assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max())), what


@form_tile
def test_b3d_geometries_against_the_oracle(ctx, form, tile, monkeypatch):
    """strided / dilated 3x3 with residual and ReLU, ragged M and N (136 channels, 12 x 16 pixels x 2 images), a 5x5 / 2 layer
    with an odd number of half steps per tap row, 1x1 with K = 1024, 48 channels (K_pad = 448: the last K step is half padding)"""
    monkeypatch.setenv("ACCEL_SPLIT", form)
    x, w, b, res = rnd(20, 2, 64, 23, 31), rnd(21, 136, 64, 3, 3, scale=0.05), rnd(22, 136), rnd(23, 2, 136, 12, 16)
    close(ctx.conv2d(x, w, b, 2, 2, 2, residual=res, act=1, tile=tile), O.relu(O.conv2d(x, w, b, 2, 2, 2) + res), "3x3 s2 d2")
    for (N, C, K, H, W, k, s, p, d) in ((2, 96, 64, 12, 10, 5, 2, 2, 1), (1, 1024, 264, 9, 13, 1, 1, 0, 1), (3, 48, 72, 20, 28, 3, 1, 1, 1),
                                       (1, 512, 136, 8, 16, 3, 1, 1, 1), (1, 16, 40, 33, 47, 3, 1, 1, 1)):
        xx, ww, bb = rnd(42, N, C, H, W), rnd(43, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(44, K)
        close(ctx.conv2d(xx, ww, bb, s, p, d, tile=tile), O.conv2d(xx, ww, bb, s, p, d), (N, C, K, k))


@form_tile
def test_b3d_is_bit_identical_to_the_register_weight_kernel(ctx, form, tile, monkeypatch):
    """M = 40960 rows: no geometry splits K, so the sums are formed in the same order"""
    monkeypatch.setenv("ACCEL_SPLIT", form)
    x, w, b = rnd(1, 1, 64, 160, 256), rnd(2, 128, 64, 1, 1, scale=0.1), rnd(3, 128)
    ref = ctx.conv2d(x, w, b, 1, 0, 1, act=1, tile=81)
    got = ctx.conv2d(x, w, b, 1, 0, 1, act=1, tile=tile)
    assert np.array_equal(got, ref)
    x3, w3 = rnd(4, 1, 32, 300, 280), rnd(5, 256, 32, 3, 3, scale=0.06)        # 329 row blocks of 256: no split-K on any geometry
    assert np.array_equal(ctx.conv2d(x3, w3, None, 1, 1, 1, tile=tile), ctx.conv2d(x3, w3, None, 1, 1, 1, tile=81))


@form_tile
def test_b3d_deconvolution_classes(ctx, form, tile, monkeypatch):
    from accel_amd import runtime
    monkeypatch.setenv("ACCEL_SPLIT", form)
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
    close(got, O.deconv2d(x[None], w, None, 2, 1))


@pytest.mark.parametrize("tile", [82, 83, 84, 85])
def test_b3d_fp16_form(ctx, tile):
    """f16 mode (plan option dtype=f16): operands rounded to half (weights on the host, pixels after the fragment read), one
    v_mfma_f32_32x32x16_f16 per fragment pair, fp32 sums -- against the oracle on half-rounded operands at fp32 rounding level"""
    os.environ["ACCEL_CONV_DTYPE"] = "f16"
    try:
        x, w, b = rnd(7, 2, 64, 40, 52), rnd(8, 136, 64, 3, 3, scale=0.05), rnd(9, 136)
        got = ctx.conv2d(x, w, b, 1, 1, 1, tile=tile)
    finally:
        os.environ.pop("ACCEL_CONV_DTYPE", None)
    h = lambda a: a.astype(np.float16).astype(np.float32)
    ref = O.conv2d(h(x), h(w), b, 1, 1, 1)
    assert float(np.abs(got - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max()))
