"""Launch geometry 78 (csrc/conv_halo.hip): the 3x3 / stride 1 / dilation 1 or 2 layers with few output channels -- the offset branches
of the deformable layers (resnet_v1_101_flownet_deeplab.py:84-130: `res5*_branch2b_offset`, 3x3, pad 2, dilate 2, 18 or 72 channels) --
in the fp16x2 form with the pixel patch staged once with its halo.  Against a float64 convolution at fp32 accumulation noise, on
ragged sizes, several images, a strided output view and a leaky activation; bit-identical with the implicit-GEMM geometry's
arithmetic is NOT required (the four wavefronts sum their taps in another order), the accuracy bar is the same."""
import numpy as np
import pytest

from accel_amd import runtime

pytestmark = pytest.mark.gpu


def conv64(x, w, d):
    """float64 3x3 stride-1 convolution at dilation d with padding d (NCHW, same-size output)"""
    N, C, H, W = x.shape
    K = w.shape[0]
    xp = np.zeros((N, C, H + 2 * d, W + 2 * d)); xp[:, :, d:d + H, d:d + W] = x
    out = np.zeros((N, K, H, W))
    w64 = w.astype(np.float64)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum('kc,nchw->nkhw', w64[:, :, ky, kx], xp[:, :, ky * d:ky * d + H, kx * d:kx * d + W])
    return out


def one_conv(ctx, N, cin, cout, H, W, d, tile, w, act=0, ycs=None, bias=None):
    m = runtime.Model(ctx)
    al = lambda b: (b + 255) // 256 * 256
    kp = (cout + 3) // 4 * 4
    ycs = ycs or kp
    o_y = al(N * H * W * cin * 4)
    m.set_param("w_weight", w)
    t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (o_y + al(N * H * W * ycs * 4), N * cin * H * W * 4, N * cout * H * W * 4)
    t += "import_nchw src=x:0:%d:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d:%d\n" % (cin, cin, H, W, N, cin, cin, H, W, N)
    t += "conv name=c in=A:0:%d:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d:%d w=w_weight act=%d slope=0.1 cin=%d cout=%d mode=conv tile=%d k=3,3 s=1,1 p=%d,%d d=%d,%d" % (
        cin, cin, H, W, N, o_y, cout, ycs, H, W, N, act, cin, cout, tile, d, d, d, d)
    if bias is not None:
        m.set_param("w_bias", bias)
        t += " bias=w_bias"
    t += "\nexport_nchw src=A:%d:%d:%d:%d:%d:%d dst=y:0:%d:%d:%d:%d:%d\n" % (o_y, cout, ycs, H, W, N, cout, cout, H, W, N)
    plan = m.add_plan("p", t)
    plan.finalize()
    return m, plan


@pytest.mark.parametrize("d", [1, 2])
@pytest.mark.parametrize("N,cin,cout,H,W,ycs,act", [
    (1, 512, 18, 64, 128, None, 0),        # res5*_branch2b_offset of the ResNet-101 trunk at the headline size
    (2, 512, 72, 32, 64, None, 0),         # the ResNet-18 trunk's, two images
    (1, 64, 18, 13, 37, 24, 2),            # ragged in both directions, strided output view, leaky activation
    (3, 96, 40, 9, 16, None, 1),           # one and a bit tile rows, a second channel strip with 8 channels in use
])
def test_halo_geometry_matches_float64(ctx, d, N, cin, cout, H, W, ycs, act):
    rng = np.random.default_rng(11 * d + cout)
    x = (np.maximum(rng.standard_normal((N, cin, H, W)), 0) * np.exp(rng.standard_normal((1, cin, 1, 1)))).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.02).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = conv64(x, w, d) + b[None, :, None, None]
    if act == 1:
        ref = np.maximum(ref, 0)
    elif act == 2:
        ref = np.where(ref > 0, ref, 0.1 * ref)
    m, plan = one_conv(ctx, N, cin, cout, H, W, d, 78, w, act=act, ycs=ycs, bias=b)
    try:
        assert [(o["tile"], o["mode"]) for o in plan.ops() if o["kind"] == "conv"] == [(78, 3)]
        m.write("x", x)
        plan.run()
        out = m.read("y", (N, cout, H, W))
        again = None
        plan.run()
        again = m.read("y", (N, cout, H, W))
    finally:
        m.close()
    s = np.abs(ref).max()
    assert np.isfinite(out).all()
    assert float(np.abs(out - ref).max() / s) <= 1e-6, float(np.abs(out - ref).max() / s)
    assert np.array_equal(out, again)          # the exchange of the four partial sums has a fixed order


def test_halo_geometry_raises_the_range_of_what_it_wrote(ctx):
    """a second fp16x2-form convolution behind it reads the slot the halo kernel's epilogue raised: the chain against float64"""
    rng = np.random.default_rng(5)
    cin, cmid, cout, H, W = 64, 32, 48, 24, 40
    x = np.maximum(rng.standard_normal((1, cin, H, W)), 0).astype(np.float32) * 37.0
    w1 = (rng.standard_normal((cmid, cin, 3, 3)) * 0.05).astype(np.float32)
    w2 = (rng.standard_normal((cout, cmid, 1, 1)) * 0.2).astype(np.float32)
    m = runtime.Model(ctx)
    try:
        al = lambda b: (b + 255) // 256 * 256
        o1 = al(H * W * cin * 4); o2 = o1 + al(H * W * cmid * 4)
        m.set_param("a_weight", w1); m.set_param("b_weight", w2)
        t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (o2 + al(H * W * cout * 4), cin * H * W * 4, cout * H * W * 4)
        t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d yr=1\n" % (cin, cin, H, W, cin, cin, H, W)
        t += "conv name=a in=A:0:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=a_weight act=1 cin=%d cout=%d mode=conv tile=78 k=3,3 s=1,1 p=2,2 d=2,2 xr=1 yr=2\n" % (
            cin, cin, H, W, o1, cmid, cmid, H, W, cin, cmid)
        t += "conv name=b in=A:%d:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=b_weight act=0 cin=%d cout=%d mode=conv tile=77 k=1,1 s=1,1 p=0,0 d=1,1 xr=2\n" % (
            o1, cmid, cmid, H, W, o2, cout, cout, H, W, cmid, cout)
        t += "export_nchw src=A:%d:%d:%d:%d:%d dst=y:0:%d:%d:%d:%d\n" % (o2, cout, cout, H, W, cout, cout, H, W)
        plan = m.add_plan("p", t)
        plan.finalize()
        m.write("x", x)
        plan.run()
        out = m.read("y", (1, cout, H, W))
        ranges = plan.ranges()
    finally:
        m.close()
    mid = np.maximum(conv64(x, w1, 2), 0)
    ref = np.einsum('kc,nchw->nkhw', w2[:, :, 0, 0].astype(np.float64), mid)
    assert float(np.abs(out - ref).max() / np.abs(ref).max()) <= 2e-6
    scale, source = ranges["b"]
    assert source == 1                                             # from the writer's epilogue, not a measuring launch
    assert 2.0 ** 13 <= scale * float(mid.max()) * (1 + 1e-6) and scale * float(mid.max()) < 2.0 ** 14 * (1 + 1e-6)
