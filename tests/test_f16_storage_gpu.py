"""Half activation storage of the fp16 mode (BASELINE config 5): arena buffers between half-capable convolutions hold half values
(accel_amd.lower.Lowering.assign_storage), read and written by the fp16 form of conv_b3d.hip alone -- pixel tiles arrive by LDS-DMA
as the MFMA fragments they are, the epilogue rounds (RTNE) AFTER scale / shift, residual and activation, a half residual is read
as half.  Specification = oracle on half-rounded operands with the stored tensors rounded once more (oracle.graphs.STORE_F16)."""
import os

import numpy as np
import pytest

from accel_amd.utils import image, synth
from oracle import graphs as G, ops as O

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def h(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _run_chain(ctx, x, w0, w1, w2, k1, s1, p1, tiles, y_half_tail=False):
    """c0: x -> r (1x1 stride s1, half out);  c1: x -> t (k1 x k1 stride s1, ReLU, half out);  c2: t -> y (1x1, + r, ReLU).
    With y_half_tail, c2 writes half as well and c3 (1x1) reads it: half in, half residual, half out in one layer."""
    from accel_amd import runtime
    N, C, H, W = x.shape
    C1, C2 = w1.shape[0], w2.shape[0]
    Ho, Wo = (H + 2 * p1 - k1) // s1 + 1, (W + 2 * p1 - k1) // s1 + 1
    al = lambda b: (b + 255) // 256 * 256
    o_x = 0
    o_r = al(N * H * W * C * 4)
    o_t = o_r + al(N * Ho * Wo * C2 * 2)
    o_u = o_t + al(N * Ho * Wo * C1 * 2)
    o_y = o_u + al(N * Ho * Wo * C2 * 2)
    tot = o_y + al(N * Ho * Wo * C2 * 4)
    m = runtime.Model(ctx)
    try:
        m.set_param("w0_weight", w0); m.set_param("w1_weight", w1); m.set_param("w2_weight", w2)
        if y_half_tail:
            m.set_param("w3_weight", np.eye(C2, dtype=np.float32).reshape(C2, C2, 1, 1))
        t = "option graph=0\noption dtype=f16\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (tot, x.nbytes, N * C2 * Ho * Wo * 4)
        sfx = ":%d" % N
        t += "import_nchw src=x:0:%d:%d:%d:%d%s dst=A:%d:%d:%d:%d:%d%s\n" % (C, C, H, W, sfx, o_x, C, C, H, W, sfx)
        t += "conv name=c0 in=A:%d:%d:%d:%d:%d%s out=A:%d:%d:%d:%d:%d:%d:h w=w0_weight act=0 cin=%d cout=%d mode=conv k=1,1 s=%d,%d p=0,0 d=1,1 tile=%d\n" % (
            o_x, C, C, H, W, sfx, o_r, C2, C2, Ho, Wo, N, C, C2, s1, s1, tiles[0])
        t += "conv name=c1 in=A:%d:%d:%d:%d:%d%s out=A:%d:%d:%d:%d:%d:%d:h w=w1_weight act=1 cin=%d cout=%d mode=conv k=%d,%d s=%d,%d p=%d,%d d=1,1 tile=%d\n" % (
            o_x, C, C, H, W, sfx, o_t, C1, C1, Ho, Wo, N, C, C1, k1, k1, s1, s1, p1, p1, tiles[1])
        c2_out = ("A:%d:%d:%d:%d:%d:%d:h" % (o_u, C2, C2, Ho, Wo, N)) if y_half_tail else ("A:%d:%d:%d:%d:%d%s" % (o_y, C2, C2, Ho, Wo, sfx))
        t += "conv name=c2 in=A:%d:%d:%d:%d:%d:%d:h out=%s res=A:%d:%d:%d:%d:%d:%d:h w=w2_weight act=1 cin=%d cout=%d mode=conv k=1,1 s=1,1 p=0,0 d=1,1 tile=%d\n" % (
            o_t, C1, C1, Ho, Wo, N, c2_out, o_r, C2, C2, Ho, Wo, N, C1, C2, tiles[2])
        if y_half_tail:
            t += "conv name=c3 in=A:%d:%d:%d:%d:%d:%d:h out=A:%d:%d:%d:%d:%d%s w=w3_weight act=0 cin=%d cout=%d mode=conv k=1,1 s=1,1 p=0,0 d=1,1 tile=%d\n" % (
                o_u, C2, C2, Ho, Wo, N, o_y, C2, C2, Ho, Wo, sfx, C2, C2, tiles[2])
        t += "export_nchw src=A:%d:%d:%d:%d:%d%s dst=y:0:%d:%d:%d:%d%s\n" % (o_y, C2, C2, Ho, Wo, sfx, C2, C2, Ho, Wo, sfx)
        plan = m.add_plan("p", t)
        m.write("x", x)
        plan.finalize()
        plan.run()
        return m.read("y", (N, C2, Ho, Wo)), [o["ksplit"] for o in plan.ops() if o["kind"] == "conv"]
    finally:
        m.close()


def _ref_chain(x, w0, w1, w2, k1, s1, p1, y_half_tail=False):
    r = h(O.conv2d(h(x), h(w0), None, s1, 0, 1))
    t = h(O.relu(O.conv2d(h(x), h(w1), None, s1, p1, 1)))
    y = O.relu(O.conv2d(t, h(w2), None, 1, 0, 1) + r)
    return h(y) if y_half_tail else y      # (c3 is the identity on half values: exact)


def _same_up_to_rounding_flips(got, ref, flips=0.05):
    """Two evaluations of a chain with ROUNDED intermediate tensors agree to fp32 rounding except where the fp32 value of an
    intermediate sits within that noise of a half-rounding boundary: there the stored half differs by one unit in the last place
    (2^-10 of its magnitude) and the outputs that read it move by weight x ulp.  So: the typical element agrees to fp32 rounding,
    few elements deviate at all, and no deviation exceeds a few half ulps of the output scale -- a wrong tile, plane, channel pair
    or residual would break all three by orders of magnitude."""
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref)
    assert float(np.median(err)) <= 2e-6 * scale, "median error %g" % float(np.median(err))
    assert float((err > 1e-5 * scale).mean()) < flips, "%.3f of the outputs deviate" % float((err > 1e-5 * scale).mean())
    assert float(err.max()) <= 4 * 2.0 ** -10 * scale, "max error %g (scale %g)" % (float(err.max()), scale)


@pytest.mark.parametrize("tiles", [(-1, -1, -1), (84, 85, 84), (89, 84, 89), (84, 84, 88)])
@pytest.mark.parametrize("k1,s1,p1", [(3, 1, 1), (3, 2, 1), (1, 2, 0)])
def test_half_views_through_a_residual_block(ctx, tiles, k1, s1, p1):
    """fp32 in -> half out, half in + half residual -> fp32 out, on every fp16 geometry of conv_b3d.hip (and on the heuristic's,
    tile -1); 136 output rows = a ragged N tile, 2 images, a pixel count that is not a multiple of the tile"""
    N, C, H, W, C1, C2 = 2, 64, 21, 38, 96, 136
    x, w0, w1, w2 = np.maximum(rnd(1, N, C, H, W), 0), rnd(2, C2, C, 1, 1, scale=0.12), rnd(3, C1, C, k1, k1, scale=(2.0 / (C * k1 * k1)) ** 0.5), rnd(4, C2, C1, 1, 1, scale=0.1)
    if 88 in tiles:
        C2 = 64
        w0, w2 = w0[:64], w2[:64]
    got, _ = _run_chain(ctx, x, w0, w1, w2, k1, s1, p1, tiles)
    ref = _ref_chain(x, w0, w1, w2, k1, s1, p1)
    _same_up_to_rounding_flips(got, ref)


@pytest.mark.parametrize("tiles", [(-1, -1, -1), (82, 83, 82)])
def test_half_in_half_residual_half_out_and_split_k(ctx, tiles):
    """all three views of one layer half; a long reduction over few pixels, so that the launches split K and the reduce kernel does
    the half residual read and the half store; 256-column tiles on 320 channels"""
    N, C, H, W, C1, C2 = 1, 512, 9, 14, 256, 320
    x, w0, w1, w2 = np.maximum(rnd(5, N, C, H, W), 0), rnd(6, C2, C, 1, 1, scale=0.04), rnd(7, C1, C, 3, 3, scale=(2.0 / (C * 9)) ** 0.5), rnd(8, C2, C1, 1, 1, scale=0.06)
    got, splits = _run_chain(ctx, x, w0, w1, w2, 3, 1, 1, tiles, y_half_tail=True)
    assert max(splits) > 1, "the shapes of this test are meant to split K (%s)" % splits
    ref = _ref_chain(x, w0, w1, w2, 3, 1, 1, y_half_tail=True)
    _same_up_to_rounding_flips(got, ref, flips=0.12)      # a K of 4608 and three rounded tensors: more values sit near a boundary


def test_only_convolutions_of_an_f16_plan_take_half_views(ctx):
    from accel_amd import runtime
    m = runtime.Model(ctx)
    try:
        m.set_param("w_weight", rnd(1, 64, 64, 1, 1))
        base = "option graph=0\n%sarena bytes=1048576\npbuf name=x bytes=%d\n" % ("%s", 64 * 64 * 4)
        conv = "conv name=c in=A:0:64:64:8:8 out=A:65536:64:64:8:8:1:h w=w_weight act=0 cin=64 cout=64 mode=conv k=1,1 s=1,1 p=0,0 d=1,1\n"
        with pytest.raises(runtime.AccelError, match="half views need an f16-mode layer"):
            m.add_plan("a", base % "" + conv).finalize()                       # fp32 plan
        pool = "pool name=q in=A:0:64:64:8:8:1:h out=A:65536:64:64:4:4 kind=max k=2,2 s=2,2 p=0,0 conv=valid\n"
        with pytest.raises(runtime.AccelError, match="only convolutions"):
            m.add_plan("b", base % "option dtype=f16\n" + pool).finalize()
    finally:
        m.close()


def half_layers(version, H, W, cfg):
    """names of the convolutions whose output the lowering stores as half (key and non-key plan of Accel-<version>)"""
    from accel_amd import lower, symbols
    inst = getattr(getattr(symbols, "accel_" + version), "accel_" + version)()
    names = set()
    for key in (True, False):
        sym = inst.get_key_test_symbol(cfg) if key else inst.get_cur_test_symbol(cfg)
        shapes = {"data": (1, 3, H, W)}
        if not key:
            shapes.update({"data_key": (1, 3, H, W), "feat_key": (1, 2048, H // 16, W // 16)})
        _, lw = lower.lower(sym, shapes, conv_dtype="f16", store_f16=True, fold_linear=False)
        names |= {a["name"] for k, a in lw.ops if k == "conv" and a["out"].buf.esize == 2}
    return names


@pytest.mark.parametrize("version", ["50", pytest.param("101", marks=pytest.mark.gpu_extra)])
def test_clip_with_half_storage_against_its_specification_512x1024(demo_cfg, monkeypatch, version):
    """Whole clip, Accel-50 (config 5's model) and Accel-101, key + non-key frame at 512x1024 in f16 mode with half storage against
    the oracle on half-rounded operands AND half-rounded stored tensors (STORE_F16 = the layers the lowering stores as half).  As
    for the fp32-storage form of the mode (tests/test_f16_gpu.py) two evaluations of a specification with ~100 discontinuous
    roundings decorrelate down to the half-precision noise, so the whole-graph assertions are statistical: the typical pixel is no
    farther from the mode's own specification than from the fp32 result, the worst pixel stays inside 10 % of the logit range,
    fewer than 0.5 % of the labels differ; the per-layer 1e-5 bar is in the tests above."""
    from accel_amd import demo
    from accel_amd.core import tester
    monkeypatch.setenv("ACCEL_CONV_DTYPE", "f16")
    monkeypatch.setenv("ACCEL_FOLD_LINEAR", "0")
    H, W, interval = 512, 1024, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        r = demo.ClipRunner(version, demo_cfg, arg, aux, (H, W))
        outs = []
        for idx, arrays in enumerate(demo.build_batches(frames, demo_cfg)):
            logits, labels = r.step(idx, arrays, interval)
            outs.append((logits.asnumpy(), np.uint8(np.squeeze(labels.asnumpy()))))
        nhalf = 0
        for pred in (r.key_predictor, r.cur_predictor):
            lw = pred.plan_for(H, W, 1)[1]
            nhalf += len(lw.half_bufs)
        assert nhalf > 100, "half storage must be on by default in f16 mode (%d half buffers)" % nhalf
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    fr = [image.transform(f, demo_cfg.network.PIXEL_MEANS).astype(np.float32) for f in frames]
    G.ROUND_F16, G.STORE_F16 = True, half_layers(version, H, W, demo_cfg)
    try:
        ref16 = G.run_clip(P, version, fr, interval)
    finally:
        G.ROUND_F16, G.STORE_F16 = False, None
    ref32 = G.run_clip(P, version, fr, interval)
    for t, ((lg, lab), (r16, l16), (r32, l32)) in enumerate(zip(outs, ref16, ref32)):
        scale = max(1.0, float(np.abs(r32).max()))
        d16, d32 = np.abs(lg - r16).ravel() / scale, np.abs(lg - r32).ravel() / scale
        q = lambda d: tuple(float(np.quantile(d[::7], p_)) for p_ in (0.5, 0.99, 0.999)) + (float(d.max()),)
        q16, q32 = q(d16), q(d32)
        print("f16 mode + half storage, accel-%s frame %d, |error| / logit range (median, 99 %%, 99.9 %%, max): vs its specification %.2e %.2e %.2e %.2e | "
              "vs the fp32 oracle %.2e %.2e %.2e %.2e; labels differing %.4f %% / %.4f %%"
              % ((version, t) + q16 + q32 + (100 * float((lab != l16[0]).mean()), 100 * float((lab != l32[0]).mean()))))
        assert q16[0] <= 1.05 * q32[0], "frame %d" % t
        assert q16[3] <= 0.1 and float((lab != l16[0]).mean()) < 5e-3, "frame %d" % t
