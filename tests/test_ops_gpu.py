"""Operator-level parity: HIP kernels (through the C ABI) vs the CPU oracle on
the same seeded inputs.  fp32; tolerance 1e-4 relative to the output scale per
op (the end-to-end logit tolerance of BASELINE.json is 1e-3)."""
import numpy as np
import pytest

from oracle import ops as O

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def close(a, b, rtol=1e-4):
    assert a.shape == b.shape, (a.shape, b.shape)
    tol = rtol * max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol, "max err %g > tol %g" % (err, tol)


CONV_CASES = [
    # C, K, H, W, k, s, p, d
    (64, 64, 24, 40, 1, 1, 0, 1),
    (64, 128, 24, 40, 3, 1, 1, 1),
    (128, 64, 25, 41, 3, 2, 1, 1),       # odd sizes, stride 2
    (32, 72, 16, 24, 3, 1, 2, 2),        # dilated offset conv, 72 channels
    (256, 18, 16, 24, 3, 1, 1, 1),       # 18-channel offset conv
    (3, 64, 64, 96, 7, 2, 3, 1),         # RGB stem
    (6, 64, 64, 96, 7, 2, 3, 1),         # FlowNet stem
    (64, 128, 32, 48, 5, 2, 2, 1),       # FlowNet conv2
    (1026, 2, 8, 16, 3, 1, 1, 1),        # flow predictor on a concat with ragged channel count
    (194, 2, 16, 32, 3, 1, 1, 1),
    (256, 512, 17, 23, 1, 2, 0, 1),      # 1x1 stride-2 shortcut
    (2048, 19, 8, 16, 1, 1, 0, 1),       # score conv
    (8, 8, 1, 1, 3, 1, 1, 1),            # degenerate 1x1 image
    (512, 1024, 4, 8, 3, 1, 1, 1),       # low-res FlowNet conv6_1 shape: split-K path
    (1024, 19, 16, 32, 1, 1, 0, 1),      # score conv: narrow N + split-K
]


@pytest.mark.parametrize("C,K,H,W,k,s,p,d", CONV_CASES)
def test_conv2d_bias(ctx, C, K, H, W, k, s, p, d):
    x, w, b = rnd(1, 1, C, H, W), rnd(2, K, C, k, k, scale=(2.0 / (C * k * k)) ** 0.5), rnd(3, K)
    close(ctx.conv2d(x, w, b, s, p, d), O.conv2d(x, w, b, s, p, d))


# 0-4: register-staged tiles, plain schedule; 5-9: software-pipelined; 10-12: 8-wave; 13/15: BK 64/16; 16-19: LDS-DMA ring
ALL_TILES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 31, 32, 33, 34, 35]


@pytest.mark.parametrize("tile", ALL_TILES)
def test_conv2d_every_tile_config(ctx, tile):
    C, K, H, W = 128, 200, 20, 36
    x, w = rnd(4, 1, C, H, W), rnd(5, K, C, 3, 3, scale=0.05)
    close(ctx.conv2d(x, w, None, 1, 1, 1, tile=tile), O.conv2d(x, w, None, 1, 1, 1))


@pytest.mark.parametrize("tile", ALL_TILES)
def test_conv2d_every_tile_config_strided_dilated_residual(ctx, tile):
    C, K, H, W = 64, 136, 23, 31          # ragged M and N tiles, borders on every side
    x, w, res = rnd(40, 1, C, H, W), rnd(41, K, C, 3, 3, scale=0.05), rnd(42, 1, K, 12, 16)
    ref = O.relu(O.conv2d(x, w, None, 2, 2, 2) + res)
    close(ctx.conv2d(x, w, None, 2, 2, 2, residual=res, act=1, tile=tile), ref)
    x1, w1 = rnd(43, 1, 256, 9, 13), rnd(44, 72, 256, 1, 1, scale=0.05)
    close(ctx.conv2d(x1, w1, None, 1, 0, 1, tile=tile), O.conv2d(x1, w1, None, 1, 0, 1))


def test_conv2d_fused_epilogue(ctx):
    C, K, H, W = 64, 96, 20, 28
    x, w = rnd(6, 1, C, H, W), rnd(7, K, C, 3, 3, scale=0.05)
    scale, shift, res = rnd(8, K), rnd(9, K), rnd(10, 1, K, H, W)
    ref = O.conv2d(x, w, None, 1, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None] + res
    close(ctx.conv2d(x, w, None, 1, 1, 1, scale=scale, shift=shift, residual=res, act=1), O.relu(ref))
    close(ctx.conv2d(x, w, None, 1, 1, 1, scale=scale, shift=shift, residual=res, act=2, slope=0.1),
          O.leaky_relu(ref, 0.1))


def test_conv2d_identity_detects_transpose(ctx):
    # A = I with an asymmetric B: a row/col swap in the MFMA C-layout cannot pass
    C = 64
    x = rnd(11, 1, C, 8, 8)
    w = np.zeros((C, C, 1, 1), np.float32)
    w[np.arange(C), (np.arange(C) * 7 + 3) % C, 0, 0] = np.arange(1, C + 1, dtype=np.float32)
    close(ctx.conv2d(x, w), O.conv2d(x, w))


@pytest.mark.parametrize("C,K,H,W,bias", [(64, 32, 9, 13, True), (1026, 512, 4, 6, True), (2, 2, 8, 16, True),
                                           (512, 256, 8, 8, False)])
def test_deconv_4x4_s2(ctx, C, K, H, W, bias):
    x, w = rnd(12, 1, C, H, W), rnd(13, C, K, 4, 4, scale=(0.5 / C) ** 0.5)
    b = rnd(14, K) if bias else None
    # pad 1 directly ...
    ref = O.deconv2d(x, w, b, 2, 1)
    close(ctx.deconv2d_4x4s2(x, w, b), ref)
    # ... equals the reference's pad 0 + Crop(offset=(1,1)) form (resnet_v1_101_flownet_deeplab.py:1775-1777)
    ref0 = O.crop_like(O.deconv2d(x, w, b, 2, 0), (2 * H, 2 * W), (1, 1))
    np.testing.assert_array_equal(ref, ref0)
    close(ctx.deconv2d_4x4s2(x, w, b, act=2, slope=0.1), O.leaky_relu(ref, 0.1))


def _offsets(seed, dg, H, W, kind):
    r = np.random.default_rng(seed)
    shp = (1, 18 * dg, H, W)
    if kind == "zero":
        return np.zeros(shp, np.float32)
    if kind == "integer":
        return r.integers(-2, 3, shp).astype(np.float32)
    if kind == "fractional":
        return r.normal(0, 1.0, shp).astype(np.float32)
    return r.normal(0, 12.0, shp).astype(np.float32)   # border-crossing stress


@pytest.mark.parametrize("dg", [1, 4])
@pytest.mark.parametrize("kind", ["zero", "integer", "fractional", "border"])
def test_deform_conv(ctx, dg, kind):
    C, K, H, W = 32, 48, 12, 20
    x, w = rnd(15, 1, C, H, W), rnd(16, K, C, 3, 3, scale=0.06)
    off = _offsets(17, dg, H, W, kind)
    ref = O.deform_conv2d(x, off, w, 1, 2, 2, dg)
    close(ctx.deform_conv2d(x, off, w, 1, 2, 2, dg), ref)
    if kind == "zero":   # zero offsets == plain dilated convolution
        close(ref, O.conv2d(x, w, None, 1, 2, 2), rtol=1e-5)


@pytest.mark.parametrize("H,W", [(32, 48), (33, 47), (7, 9)])
@pytest.mark.parametrize("kind,k,s,p,conv", [("max", 3, 2, 0, "full"), ("max", 3, 2, 1, "valid"), ("avg", 2, 2, 0, "full")])
def test_pool(ctx, H, W, kind, k, s, p, conv):
    x = rnd(18, 1, 20, H, W)
    got, ref = ctx.pool2d(x, kind, k, s, p, conv), O.pool2d(x, kind, k, s, p, conv)
    if kind == "max":
        np.testing.assert_array_equal(got, ref)
    else:
        close(got, ref, 1e-6)


def test_pool_bn_relu_epilogue(ctx):
    x, scale, shift = rnd(19, 1, 64, 16, 24), rnd(20, 64), rnd(21, 64)
    ref = O.relu(O.pool2d(x, "max", 3, 2, 1, "valid") * scale[None, :, None, None] + shift[None, :, None, None])
    close(ctx.pool2d(x, "max", 3, 2, 1, "valid", scale=scale, shift=shift, relu=True), ref, 1e-6)


@pytest.mark.parametrize("mag", [0.0, 0.7, 3.0, 40.0])
def test_flow_warp(ctx, mag):
    C, H, W = 64, 12, 20
    feat, flow = rnd(22, 1, C, H, W), rnd(23, 1, 2, H, W, scale=mag)
    close(ctx.flow_warp(feat, flow), O.flow_warp(feat, flow), 1e-5)
    if mag == 0.0:
        close(O.flow_warp(feat, flow), feat, 1e-5)


def test_flow_warp_reproduces_the_mxnet_docstring_example(ctx):
    """the worked example of MXNet's BilinearSampler documentation for the `warp` grid (tests/test_pins_cpu.py holds the same vector for the
    oracle): flow (1, 0) shifts the image one pixel to the left, zeros come in from outside -- exact on the device too"""
    data = np.array([[[[1, 4, 3, 6], [1, 8, 8, 9], [0, 4, 1, 5], [1, 0, 1, 3]]]], np.float32)
    want = np.array([[[[4, 3, 6, 0], [8, 8, 9, 0], [4, 1, 5, 0], [0, 1, 3, 0]]]], np.float32)
    flow = np.zeros((1, 2, 4, 4), np.float32)
    flow[:, 0] = 1.0
    got = ctx.flow_warp(np.repeat(data, 4, axis=1), flow)      # the kernel moves channel quads
    assert np.array_equal(got, np.repeat(want, 4, axis=1))


def test_flow_input(ctx):
    cur, prev = rnd(24, 1, 3, 32, 64, scale=60), rnd(25, 1, 3, 32, 64, scale=60)
    data = np.concatenate([cur / np.float32(255.0), prev / np.float32(255.0)], axis=1)
    close(ctx.flow_input(cur, prev), O.pool2d(data, "avg", 2, 2, 0, "full"), 1e-6)


def _tail_ref(left, wl, right=None, wr=None, cw=None, cb=None):
    n = left.shape[1]
    H, W = left.shape[2] * 16, left.shape[3] * 16
    a = O.crop_like(O.deconv2d(left, wl, None, 16, 0, groups=n), (H, W), (8, 8))
    if right is None:
        return a
    b = O.crop_like(O.deconv2d(right, wr, None, 16, 0, groups=n), (H, W), (8, 8))
    return O.conv2d(np.concatenate([a, b], axis=1), cw, cb)


@pytest.mark.parametrize("bilinear", [True, False])
@pytest.mark.parametrize("two", [False, True])
def test_score_fuse(ctx, bilinear, two):
    from accel_amd.utils.synth import bilinear_kernel
    n, Hs, Ws = 19, 4, 6
    left, right = rnd(26, 1, n, Hs, Ws, scale=3), rnd(27, 1, n, Hs, Ws, scale=3)
    wl = bilinear_kernel(n, 32) if bilinear else rnd(28, n, 1, 32, 32, scale=0.2)
    wr = bilinear_kernel(n, 32) if bilinear else rnd(29, n, 1, 32, 32, scale=0.2)
    cw, cb = rnd(30, n, 2 * n, 1, 1, scale=0.3), rnd(31, n)
    if two:
        logits, labels = ctx.score_fuse(left, wl, right, wr, cw, cb)
        ref = _tail_ref(left, wl, right, wr, cw, cb)
    else:
        logits, labels = ctx.score_fuse(left, wl)
        ref = _tail_ref(left, wl)
    close(logits, ref, 1e-5)
    np.testing.assert_array_equal(labels, O.argmax_c(logits))   # fused argmax == argmax of its own logits
    srt = np.sort(ref, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3
    np.testing.assert_array_equal(labels[safe], O.argmax_c(ref)[safe])


def test_argmax_ties_pick_first(ctx):
    x = np.zeros((1, 19, 4, 8), np.float32)
    x[0, 5] = 1.0
    x[0, 9] = 1.0          # exact tie -> first index (5)
    x[0, 3, 0, 0] = 2.0
    got = ctx.argmax_c(x)
    np.testing.assert_array_equal(got, O.argmax_c(x))
    assert got[0, 0, 0] == 3 and got[0, 1, 1] == 5


# ---- batched operators: images stacked pixel-major, M = N*Ho*Wo inside the conv kernel ---------------------
@pytest.mark.parametrize("tile", [-1, 8, 10, 12, 31, 32, 33, 34, 9])
def test_conv2d_batched(ctx, tile):
    N, C, K, H, W = 3, 64, 136, 19, 27                   # ragged tiles; M = 3*10*14 rows straddle image boundaries
    x, w, b = rnd(70, N, C, H, W), rnd(71, K, C, 3, 3, scale=0.05), rnd(72, K)
    res = rnd(73, N, K, 10, 14)
    ref = O.relu(O.conv2d(x, w, b, 2, 1, 1) + res)
    close(ctx.conv2d(x, w, b, 2, 1, 1, residual=res, act=1, tile=tile), ref)


def test_conv2d_batched_split_k_and_narrow(ctx):
    x, w = rnd(74, 2, 512, 4, 8), rnd(75, 1024, 512, 3, 3, scale=0.03)      # low-res FlowNet shape: split-K path
    close(ctx.conv2d(x, w, None, 1, 1, 1), O.conv2d(x, w, None, 1, 1, 1))
    x2, w2, b2 = rnd(76, 3, 196, 12, 16), rnd(77, 2, 196, 3, 3, scale=0.05), rnd(78, 2)   # flow predictor: narrow-N kernel
    close(ctx.conv2d(x2, w2, b2, 1, 1, 1), O.conv2d(x2, w2, b2, 1, 1, 1))


@pytest.mark.parametrize("N,C,K,H,W", [
    (1, 388, 2, 13, 21),        # strips of 4 with a ragged last strip, two channel passes (97 quads)
    (2, 196, 2, 64, 128),       # big enough for strips of 8
    (3, 1028, 2, 40, 100),      # strips of 8, width not a multiple of 8, 5 channel passes
    (2, 772, 1, 9, 7),          # one real output channel, map narrower than a strip
    (1, 40, 2, 3, 3),           # fewer channel quads than lanes
])
def test_flow_predictor_strip_kernel(ctx, N, C, K, H, W):
    """3x3 / stride 1 / pad 1 convolutions with <= 2 output channels take the strip kernel (conv_narrow3x3_kernel):
    borders on every side, ragged strips, batch, and the fused epilogue."""
    x, w, b = rnd(90, N, C, H, W), rnd(91, K, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5), rnd(92, K)
    close(ctx.conv2d(x, w, b, 1, 1, 1), O.conv2d(x, w, b, 1, 1, 1))
    scale, shift, res = rnd(93, K), rnd(94, K), rnd(95, N, K, H, W)
    ref = O.conv2d(x, w, None, 1, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None] + res
    close(ctx.conv2d(x, w, None, 1, 1, 1, scale=scale, shift=shift, residual=res, act=2, slope=0.1), O.leaky_relu(ref, 0.1))


def test_deconv_deform_pool_batched(ctx):
    x, w = rnd(80, 2, 36, 6, 9), rnd(81, 36, 24, 4, 4, scale=0.1)
    close(ctx.deconv2d_4x4s2(x, w), O.deconv2d(x, w, None, stride=2, pad=1))
    xd, wd = rnd(82, 2, 16, 9, 11), rnd(83, 24, 16, 3, 3, scale=0.1)
    off = rnd(84, 2, 18, 9, 11, scale=1.5)
    close(ctx.deform_conv2d(xd, off, wd, 1, 2, 2, 1), O.deform_conv2d(xd, off, wd, 1, 2, 2, 1))
    xp = rnd(85, 3, 12, 17, 23)
    close(ctx.pool2d(xp, "max", 3, 2, 1, convention="full"), O.pool2d(xp, "max", 3, 2, 1, convention="full"))


def _fuzz_cases(n, seed=20260929):
    rng = np.random.RandomState(seed)
    tiles = [-1, -1, 3, 8, 10, 11, 12, 31, 32, 33, 34, 4, 9]
    out = []
    for i in range(n):
        k = int(rng.choice([1, 3, 3, 5, 7]))
        d = int(rng.choice([1, 1, 2])) if k > 1 else 1
        s = int(rng.choice([1, 1, 2]))
        pad = int(rng.randint(0, (k // 2) * d + 1))
        C = int(rng.choice([3, 6, 8, 20, 32, 64, 96, 130, 256]))
        K = int(rng.choice([2, 5, 19, 32, 64, 72, 100, 136, 260]))
        N = int(rng.choice([1, 1, 2, 3]))
        H, W = int(rng.randint(d * (k - 1) + 1, 34)), int(rng.randint(d * (k - 1) + 1, 40))
        out.append((i, N, C, K, H, W, k, s, pad, d, bool(rng.randint(2)), int(rng.randint(3)), bool(rng.randint(2)), int(rng.choice(tiles))))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(48), ids=lambda c: "n%d_c%d_k%d_%dx%d_k%ds%dp%dd%d_t%d" % (c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[13]))
def test_conv2d_fuzz(ctx, case):
    """Seeded random geometries (channels off the 4/32 granules, every kernel/stride/dilation/padding mix the path
    uses, batches, bias / residual / activation, forced and heuristic tiles) against the oracle."""
    i, N, C, K, H, W, k, s, pad, d, use_bias, act, use_res, tile = case
    x, w = rnd(1000 + i, N, C, H, W), rnd(2000 + i, K, C, k, k, scale=1.0 / np.sqrt(C * k * k))
    b = rnd(3000 + i, K) if use_bias else None
    ref = O.conv2d(x, w, b, s, pad, d)
    res = rnd(4000 + i, *ref.shape) if use_res else None
    if res is not None:
        ref = ref + res
    if act == 1:
        ref = O.relu(ref)
    elif act == 2:
        ref = np.where(ref > 0, ref, 0.1 * ref)
    if K <= 4 and tile in (10, 11, 12, 32, 33, 34):
        tile = -1
    got = ctx.conv2d(x, w, b, s, pad, d, residual=res, act=act, tile=tile)
    close(got, ref)


# ---- Winograd F(2x2,3x3) kernel (launch geometry id 40, conv_wino.hip): 3x3 / stride 1 / dilation 1 / pad 1 ----------
WINO_CASES = [
    # N, C, K, H, W
    (1, 64, 64, 16, 32),        # exactly one tile block / one channel block
    (1, 256, 256, 8, 16),       # res4 branch2b shape (small image), 4 channel blocks, long K loop
    (1, 128, 200, 20, 36),      # ragged channel block (200 = 3 x 64 + 8), ragged tile block (180 tiles)
    (3, 64, 72, 10, 14),        # batch: tile blocks straddle images; 72 channels
    (1, 16, 40, 6, 6),          # shortest K loop (2 steps), 9 tiles
    (2, 512, 64, 4, 8),         # deep K, low resolution
    (1, 64, 18, 12, 20),        # 18-channel offset conv shape
    (2, 64, 128, 128, 256),     # 512 blocks: the un-split path (smaller grids split K over blockIdx.y, reduce kernel finishes)
]


# 41 = the same position-GEMMs on the bf16 matrix cores with three exact bf16 terms per operand (conv_wino_b3.hip);
# 42 = 41 with rectangular tile blocks whose patch union is loaded once into an LDS copy; 43 = 42 with half the block, two per CU
@pytest.mark.parametrize("wt", [40, 41, 42, 43])
@pytest.mark.parametrize("N,C,K,H,W", WINO_CASES)
def test_conv2d_winograd_matches_oracle(ctx, N, C, K, H, W, wt):
    x, w, b = rnd(50, N, C, H, W), rnd(51, K, C, 3, 3, scale=(2.0 / (C * 9)) ** 0.5), rnd(52, K)
    ref = O.conv2d(x, w, b, 1, 1, 1)
    close(ctx.conv2d(x, w, b, 1, 1, 1, tile=wt), ref)
    # and against the direct kernel on the same inputs: two evaluations of one function on this GPU
    direct = ctx.conv2d(x, w, b, 1, 1, 1, tile=3)
    close(ctx.conv2d(x, w, b, 1, 1, 1, tile=wt), direct, 2e-5)


def test_conv2d_winograd_b3_equals_the_fp32_winograd_kernel_to_rounding(ctx):
    """Geometry 41 evaluates the SAME expression as geometry 40 (fp32 B^T d B, fp32 U, fp32 sums) with the products formed
    from three-term splits: the two must agree to fp32 rounding of the sums, far inside the op tolerance."""
    for (N, C, K, H, W) in ((1, 256, 256, 16, 32), (2, 64, 64, 32, 32), (1, 512, 128, 8, 16)):
        x, w, b = rnd(53, N, C, H, W, scale=3.0), rnd(54, K, C, 3, 3, scale=(2.0 / (C * 9)) ** 0.5), rnd(55, K)
        a, bb, cc = ctx.conv2d(x, w, b, 1, 1, 1, tile=40), ctx.conv2d(x, w, b, 1, 1, 1, tile=41), ctx.conv2d(x, w, b, 1, 1, 1, tile=42)
        dd = ctx.conv2d(x, w, b, 1, 1, 1, tile=43)
        assert float(np.abs(bb - dd).max()) <= 2e-6 * max(1.0, float(np.abs(a).max())), (N, C, K)
        assert float(np.abs(a - bb).max()) <= 4e-6 * max(1.0, float(np.abs(a).max())), (N, C, K)
        # 41 and 42 differ only in how the patches reach the transform (and possibly in the split-K factor)
        assert float(np.abs(bb - cc).max()) <= 2e-6 * max(1.0, float(np.abs(a).max())), (N, C, K)


@pytest.mark.parametrize("wt", [40, 41, 42, 43])
def test_conv2d_winograd_fused_epilogue_and_borders(ctx, wt):
    C, K, H, W = 64, 96, 20, 28
    x, w = rnd(60, 1, C, H, W), rnd(61, K, C, 3, 3, scale=0.05)
    scale, shift, res = rnd(62, K), rnd(63, K), rnd(64, 1, K, H, W)
    ref = O.conv2d(x, w, None, 1, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None] + res
    close(ctx.conv2d(x, w, None, 1, 1, 1, scale=scale, shift=shift, residual=res, act=1, tile=wt), O.relu(ref))
    close(ctx.conv2d(x, w, None, 1, 1, 1, scale=scale, shift=shift, residual=res, act=2, slope=0.1, tile=wt),
          O.leaky_relu(ref, 0.1))
    # an impulse in every corner and on every edge: the zero padding of the 4x4 input patches
    xi = np.zeros((1, 16, 6, 8), np.float32)
    for (yy, xx) in ((0, 0), (0, 7), (5, 0), (5, 7), (0, 3), (5, 4), (2, 0), (3, 7)):
        xi[0, :, yy, xx] = rnd(65 + yy + xx, 16)
    wi = rnd(66, 32, 16, 3, 3)
    np.testing.assert_allclose(ctx.conv2d(xi, wi, None, 1, 1, 1, tile=wt), O.conv2d(xi, wi, None, 1, 1, 1), rtol=0, atol=2e-5)


def test_conv2d_winograd_rejects_other_geometries(ctx):
    from accel_amd.runtime import AccelError
    x, w = rnd(70, 1, 64, 8, 8), rnd(71, 64, 64, 3, 3)
    for (s, p, d) in ((2, 1, 1), (1, 2, 2), (1, 0, 1)):
        with pytest.raises(AccelError, match="Winograd"):
            ctx.conv2d(x, w, None, s, p, d, tile=40)
    with pytest.raises(AccelError, match="Winograd"):
        ctx.conv2d(rnd(72, 1, 64, 7, 9), w, None, 1, 1, 1, tile=40)       # odd output size
    with pytest.raises(AccelError, match="Winograd"):
        ctx.conv2d(x, w, None, 2, 1, 1, tile=41)
    with pytest.raises(AccelError, match="Winograd"):
        ctx.conv2d(rnd(73, 1, 24, 8, 8), rnd(74, 64, 24, 3, 3), None, 1, 1, 1, tile=41)     # 41 needs channels in multiples of 16
    with pytest.raises(AccelError, match="not part of this build"):
        ctx.conv2d(x, w, None, 1, 1, 1, tile=21)                          # timing-only ablation id: diagnostics build only


# ---- direct 7x7/2 stem kernel (launch geometry id 50, conv_stem.hip) and the same layer on the bf16 matrix cores, three exact bf16
# ---- terms per operand (id 51, conv_stem_b3.hip) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [50, 51])
@pytest.mark.parametrize("N,H,W", [(1, 64, 96), (2, 50, 70), (1, 16, 128), (3, 37, 131), (1, 128, 256), (3, 250, 518)])
def test_conv2d_stem_kernel_matches_oracle(ctx, tile, N, H, W):
    """one tile, ragged tiles in both directions, and (last case: 3 x 125 x 259 outputs = 240 tiles of 8 x 64) more tiles than a
    quarter of the persistent blocks so that blocks walk on and both window stages get reused"""
    x, w, b = rnd(80, N, 3, H, W, scale=50.0), rnd(81, 64, 3, 7, 7, scale=(2.0 / 147) ** 0.5 / 50.0), rnd(82, 64)
    ref = O.conv2d(x, w, b, 2, 3, 1)
    close(ctx.conv2d(x, w, b, 2, 3, 1, tile=tile), ref)
    scale, shift = rnd(83, 64), rnd(84, 64)
    ref2 = O.relu(O.conv2d(x, w, None, 2, 3, 1) * scale[None, :, None, None] + shift[None, :, None, None])
    close(ctx.conv2d(x, w, None, 2, 3, 1, scale=scale, shift=shift, act=1, tile=tile), ref2)


def test_conv2d_stem_bf16x3_error_is_of_the_order_of_the_fp32_kernels(ctx):
    """against a float64 convolution: the three-term split adds nothing measurable to the fp32 accumulation noise"""
    import torch
    import torch.nn.functional as F
    x, w = rnd(90, 2, 3, 120, 200, scale=50.0), rnd(91, 64, 3, 7, 7, scale=0.002)
    truth = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride=2, padding=3).numpy()
    sc = max(1.0, float(np.abs(truth).max()))
    e50 = float(np.abs(ctx.conv2d(x, w, None, 2, 3, 1, tile=50) - truth).max()) / sc
    e51 = float(np.abs(ctx.conv2d(x, w, None, 2, 3, 1, tile=51) - truth).max()) / sc
    print("stem, max error / output scale vs float64: fp32 MFMA %.2e, bf16x3 %.2e" % (e50, e51))
    assert e51 <= 3e-6 and e51 <= 4 * e50 + 2e-7


@pytest.mark.parametrize("tile", [50, 51])
def test_conv2d_stem_kernel_rejects_other_layers(ctx, tile):
    from accel_amd.runtime import AccelError
    with pytest.raises(AccelError, match="stem kernel"):
        ctx.conv2d(rnd(85, 1, 6, 32, 32), rnd(86, 64, 6, 7, 7), None, 2, 3, 1, tile=tile)       # FlowNet's 6-channel stem
    with pytest.raises(AccelError, match="stem kernel"):
        ctx.conv2d(rnd(87, 1, 3, 32, 32), rnd(88, 32, 3, 7, 7), None, 2, 3, 1, tile=tile)       # 32 output channels


# ---- weight-stationary streaming 1x1 kernel (launch geometry id 60, conv_1x1ws.hip) --------------------------------------
@pytest.mark.parametrize("N,C,K,H,W", [
    (1, 64, 256, 32, 48),        # res2 branch1 / branch2c shape class: 12 pixel tiles of 128, fewer tiles than CUs
    (2, 64, 256, 37, 53),        # ragged last pixel tile, batch
    (1, 64, 512, 24, 40),        # two column groups of 256 channels
    (1, 128, 512, 40, 72),       # res3 branch2c class: pixel tiles of 64, four column groups of 128
    (3, 128, 128, 19, 23),       # one column group, ragged
    (1, 64, 256, 160, 288),      # more pixel tiles than persistent blocks: every block streams several
])
def test_conv2d_weight_stationary_1x1_matches_oracle(ctx, N, C, K, H, W):
    x, w, b = rnd(100, N, C, H, W), rnd(101, K, C, 1, 1, scale=(2.0 / C) ** 0.5), rnd(102, K)
    close(ctx.conv2d(x, w, b, 1, 0, 1, tile=60), O.conv2d(x, w, b, 1, 0, 1))
    scale, shift, res = rnd(103, K), rnd(104, K), rnd(105, N, K, H, W)
    ref = O.conv2d(x, w, None, 1, 0, 1) * scale[None, :, None, None] + shift[None, :, None, None] + res
    close(ctx.conv2d(x, w, None, 1, 0, 1, scale=scale, shift=shift, residual=res, act=1, tile=60), O.relu(ref))
    close(ctx.conv2d(x, w, None, 1, 0, 1, scale=scale, shift=shift, residual=res, tile=60), ref)


def test_conv2d_weight_stationary_rejects_other_layers(ctx):
    from accel_amd.runtime import AccelError
    for args in ((rnd(106, 1, 256, 8, 8), rnd(107, 64, 256, 1, 1), 1, 0),        # K = 256
                 (rnd(108, 1, 64, 8, 8), rnd(109, 128, 64, 1, 1), 1, 0),         # 128 output channels from 64
                 (rnd(110, 1, 64, 8, 8), rnd(111, 256, 64, 1, 1), 2, 0),         # stride 2
                 (rnd(112, 1, 64, 8, 8), rnd(113, 256, 64, 3, 3), 1, 1)):        # 3x3
        with pytest.raises(AccelError, match="weight-stationary"):
            ctx.conv2d(args[0], args[1], None, args[2], args[3], 1, tile=60)
    with pytest.raises(AccelError, match="weight-stationary"):
        ctx.conv2d(rnd(114, 1, 64, 8, 8), rnd(115, 256, 64, 1, 1), None, 1, 0, 1, act=2, slope=0.1, tile=60)     # leaky ReLU
