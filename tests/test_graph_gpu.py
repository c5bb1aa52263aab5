"""Whole-graph parity on the GPU: the demo.py schedule (key frame, then
non-key frames chained through the propagated feature) on the HIP path vs the
CPU oracle, same seeded weights and frames.

Tolerance: |logit - oracle| <= 1e-3 * max(1, max|oracle logit|)  (BASELINE.json:
"logits within 1e-3 fp32"); label maps must be identical wherever the oracle's
top-2 margin exceeds twice the MEASURED logit error of the frame (inside that band
a tie can legitimately flip), and the mismatch fraction overall must stay below
0.1 %.  tests/parity_report.py logs the margin histogram of every frame."""
import os

import numpy as np
import pytest

from accel_amd.utils import image, synth
from oracle import graphs as G

from parity_report import check_against_oracle

pytestmark = pytest.mark.gpu


def _oracle_frames(frames_bgr, cfg):
    return [image.transform(f, cfg.network.PIXEL_MEANS).astype(np.float32) for f in frames_bgr]


def _check(outs, ref, tag):
    """logits within 1e-3 of the oracle; labels identical wherever the oracle's top-2 margin exceeds twice the MEASURED
    logit error; margin histogram logged (tests/parity_report.py)"""
    check_against_oracle(outs, ref, tag)


@pytest.mark.parametrize("version", ["18", pytest.param("34", marks=pytest.mark.gpu_extra)])
def test_clip_parity_without_linear_fold(demo_cfg, version, monkeypatch):
    """ACCEL_FOLD_LINEAR=0 runs `feat_upsampling` and `fc6` as the reference's two separate layers; the default
    (composed deconvolution) is what every other test in this file exercises.  Both must match the oracle."""
    from accel_amd import demo
    from accel_amd.core import tester
    monkeypatch.setenv("ACCEL_FOLD_LINEAR", "0")
    H, W, interval = 128, 256, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        runner_outs = demo.run_clip(version, demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, version, _oracle_frames(frames, demo_cfg), interval)
    _check(runner_outs, ref, "accel-%s unfolded" % version)


@pytest.mark.parametrize("version", ["18", "101"])
def test_clip_parity_dcn_stress_offsets_x20(demo_cfg, version):
    """SURVEY.md 8d "stress set x20": the deformable layers' offset convolutions drawn 20x wider (offsets of 15-30 px, most
    taps of the outer rings land outside the image), whole clip against the oracle.  The operator is discontinuous at the
    border, so isolated footprints may deviate -- each one must be VERIFIED against the border taps the oracle recorded
    (parity_report.flip_windows); everything else meets the usual tolerance."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 256, 512, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg, offset_std=0.032)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip(version, demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, version, _oracle_frames(frames, demo_cfg), interval)
    npts = [len(c) for c in ref.critical]
    print("dcn stress accel-%s: oracle border-tap points per frame %s" % (version, npts))
    check_against_oracle(outs, ref, "accel-%s dcn-stress" % version, max_windows=6)


@pytest.mark.parametrize("version", ["18", "34", "50", "101"])
def test_clip_parity_128x256(demo_cfg, version):
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 128, 256, 3
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    frames = synth.make_clip(H, W, 4)        # key, cur, cur, key
    try:
        outs = demo.run_clip(version, demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, version, _oracle_frames(frames, demo_cfg), interval)
    _check(outs, ref, "accel-" + version)


def test_feat_handle_roundtrip(demo_cfg):
    """feat returned by im_segment is a device handle; fetching it and feeding the
    host copy back must give the same result as feeding the handle (alias-safe copy-in,
    DataParallelExecutorGroup.py:18-27)."""
    from accel_amd import demo, mx
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    data = demo.build_batches(frames, demo_cfg)
    try:
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        r.step(0, data[0], 5)
        feat_host = r.feat.asnumpy().copy()
        assert feat_host.shape == (1, 2048, H // 16, W // 16)
        lg1, _ = r.step(1, data[1], 5)
        a = lg1.asnumpy().copy()
        r.step(0, data[0], 5)
        r.feat = mx.nd.array(feat_host)       # host array instead of the HBM handle
        lg2, _ = r.step(1, data[1], 5)
        # the host upload makes the derived buffer featG (= fc6_weight * feat) stale: the cur plan rebuilds it with
        # the init:featG plan, a separately tuned launch of the same conv -> equal up to summation order
        b = lg2.asnumpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * max(1.0, float(np.abs(a).max())))
    finally:
        tester.release_models()


def test_stale_derived_buffer_without_init_plan_is_an_error(demo_cfg):
    """A non-key plan that reads featG while it is stale (nothing produced it, no `init:featG` plan registered) must
    refuse to run rather than warp garbage; registering the init plan makes the same call succeed."""
    from accel_amd import lower, runtime
    from accel_amd.symbols.accel_18 import accel_18
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    sym = accel_18().get_cur_test_symbol(demo_cfg)
    shapes = {"data": (1, 3, H, W), "data_key": (1, 3, H, W), "feat_key": (1, 2048, H // 16, W // 16)}
    text, lw = lower.lower(sym, shapes)
    assert list(lw.derived_bufs) == ["featG"]
    ctx = runtime.Context(0)
    m = runtime.Model(ctx)
    try:
        m.set_params(arg, aux)
        for name, w in lower.fold_params(lw.derived, arg).items():
            m.set_param(name, w)
        plan = m.add_plan("cur", text)
        plan.finalize()
        with pytest.raises(runtime.AccelError, match="init:featG"):
            plan.run()
        m.add_plan("init:featG", lower.init_plan_text("featG", lw.derived_bufs["featG"])).finalize()
        plan.run()
        plan.run()          # featG is now kept valid by the plan itself
        ctx.sync()
    finally:
        m.close()
        ctx.close()


def test_missing_param_raises(demo_cfg):
    from accel_amd import demo
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    del arg["fc6_weight"]
    try:
        with pytest.raises(RuntimeError, match="fc6_weight not initialized"):
            demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
    finally:
        tester.release_models()


def test_dff_only_clip(demo_cfg):
    """README row "DFF": key frames + flow propagation, no correction branch."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 128, 256, 3
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("dff", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 3)
    try:
        outs = demo.run_clip("dff", demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    _check(outs, G.run_clip(P, "dff", _oracle_frames(frames, demo_cfg), interval), "dff")


def test_deeplab_frame_by_frame_baseline(demo_cfg):
    """deeplab/ test symbol: `data` only, `softmax_output` (key graph + softmax)."""
    from accel_amd import mx
    from accel_amd.core import tester
    from accel_amd.symbols.resnet_v1_101_deeplab_dcn import resnet_v1_101_deeplab_dcn
    H, W = 128, 256
    inst = resnet_v1_101_deeplab_dcn()
    sym = inst.get_symbol(demo_cfg, is_train=False)
    assert sym.list_outputs() == ["softmax_output"]
    inst.infer_shape({"data": (1, 3, H, W)})
    arg, aux = synth.make_params(inst.arg_shape_dict, inst.aux_shape_dict, data_names=("data", "softmax_label"))
    frame = image.transform(synth.make_clip(H, W, 1)[0], demo_cfg.network.PIXEL_MEANS).astype(np.float32)
    try:
        pred = tester.Predictor(sym, ["data"], ["softmax_label"], context=[mx.gpu(0)],
                                provide_data=[[("data", (1, 3, H, W))]], provide_label=[None],
                                arg_params=arg, aux_params=aux)
        out = pred.predict(mx.io.DataBatch(data=[[mx.nd.array(frame)]], label=[], provide_data=[[("data", frame.shape)]]))[0]
        prob = out["softmax_output"].asnumpy()
        lab = mx.nd.argmax(out["softmax_output"], axis=1).asnumpy()
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.deeplab_forward(P, frame)["softmax_output"]
    assert float(np.abs(prob - ref).max()) <= 1e-4          # probabilities are in [0, 1]
    np.testing.assert_allclose(prob.sum(axis=1), 1.0, atol=1e-5)
    srt = np.sort(ref, axis=1)
    safe = ((srt[:, -1] - srt[:, -2]) > 1e-3)[0]
    np.testing.assert_array_equal(lab[0][safe], np.argmax(ref, axis=1)[0][safe])


@pytest.mark.parametrize("H,W", [(256, 384), pytest.param(384, 128, marks=pytest.mark.gpu_extra), pytest.param(160, 288, marks=pytest.mark.gpu_extra), pytest.param(96, 224, marks=pytest.mark.gpu_extra)])
def test_other_aspect_ratios(demo_cfg, H, W):
    """sizes other than 1:2, and sizes that are multiples of 32 but not of 128 (odd FlowNet encoder sizes: the
    decoder's Crop(offset 1) then keeps 2h-1 rows of a deconvolution, resnet_v1_101_flownet_deeplab.py:1776-1801);
    key + one non-key frame vs the oracle"""
    from accel_amd import demo
    from accel_amd.core import tester
    demo_cfg.SCALES[0] = (min(H, W), max(H, W))     # (short-side target, long-side cap), lib/utils/image.py:194-205
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip("18", demo_cfg, arg, aux, frames, 2)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    _check(outs, G.run_clip(P, "18", _oracle_frames(frames, demo_cfg), 2), "accel-18 %dx%d" % (H, W))


@pytest.mark.parametrize("version,H,W", [("18", 144, 272), ("34", 208, 176), ("101", 144, 272), pytest.param("18", 1040, 2064, marks=pytest.mark.gpu_extra)])
def test_sizes_that_are_multiples_of_16_only(demo_cfg, version, H, W):
    """The reference binds any size its own shape inference accepts, i.e. any multiple of 16 (the head upsamples H/16 x W/16
    by exactly 16).  At multiples of 16 that are not multiples of 32 the stride-32 correction branch of Accel-18 / 34,
    upsampled 2x, is one row / column LARGER than the stride-16 map (9 vs 10 rows at H = 144): Deconvolution 32x32/16 +
    Crop(8, 8) never reaches the extra row; the fused score tail takes the two maps at their own sizes.  Odd sizes also
    run through the 'full' / 'valid' pooling conventions and the 2h-1 crops of the FlowNet decoder."""
    from accel_amd import demo
    from accel_amd.core import tester
    demo_cfg.SCALES[0] = (min(H, W), max(H, W))
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip(version, demo_cfg, arg, aux, frames, 2)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    _check(outs, G.run_clip(P, version, _oracle_frames(frames, demo_cfg), 2), "accel-%s %dx%d" % (version, H, W))


def test_size_not_multiple_of_16_is_rejected(demo_cfg):
    from accel_amd import demo
    from accel_amd.core import tester
    demo_cfg.SCALES[0] = (200, 256)
    arg, aux = synth.model_params("18", 256, 256, demo_cfg)
    try:
        with pytest.raises(ValueError, match="multiples of 16"):
            demo.ClipRunner("18", demo_cfg, arg, aux, (200, 256))
    finally:
        tester.release_models()


def test_two_resolutions_in_one_process(demo_cfg, monkeypatch):
    """a second frame size re-lowers and binds its own model (MutableModule rebinds on a shape change,
    module.py:1026-1042); results of the first size are unaffected, and the lazily bound size is CORRECT from its first
    non-key frame on: the cur plan of the new size is finalized (autotuned, warmed up, captured) between the key frame
    and the first non-key frame of the clip, which must not disturb the propagated feature."""
    from accel_amd import demo
    from accel_amd.core import tester
    monkeypatch.setenv("ACCEL_AUTOTUNE", "0")      # (a binding test: the static launch geometries; timing 256x256's layers took 20 s of the suite)
    arg, aux = synth.model_params("18", 128, 256, demo_cfg)
    P = dict(arg)
    P.update(aux)
    try:
        outs = {}
        for (H, W) in ((128, 256), (256, 256), (128, 256)):
            demo_cfg.SCALES[0] = (min(H, W), max(H, W))
            frames = synth.make_clip(H, W, 2)
            data = demo.build_batches(frames, demo_cfg)
            r = outs.setdefault("runner", demo.ClipRunner("18", demo_cfg, arg, aux, (128, 256)))
            res = []
            for i in range(2):
                lg, lab = r.step(i, data[i], 2)
                res.append((lg.asnumpy().copy(), np.uint8(lab.asnumpy()[0])))
            outs.setdefault((H, W), []).append(res)
            if (H, W) == (256, 256):
                _check(res, G.run_clip(P, "18", _oracle_frames(frames, demo_cfg), 2), "accel-18 lazily bound 256x256")
        a, b = outs[(128, 256)]
        np.testing.assert_array_equal(a[1][0], b[1][0])
        assert outs[(256, 256)][0][1][0].shape == (1, 19, 256, 256)
    finally:
        tester.release_models()


def test_c_abi_frame_entry_points(demo_cfg):
    """accel_key_forward / accel_cur_forward (include/accel_hip.h): host buffers in, NCHW feature,
    logits and uint8 labels out -- same numbers as the Predictor route on the same model."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = _oracle_frames(synth.make_clip(H, W, 2), demo_cfg)
    data = demo.build_batches(synth.make_clip(H, W, 2), demo_cfg)
    try:
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        model = r.key_predictor._model
        lg0, lab0 = r.step(0, data[0], 2)
        ref_key = (lg0.asnumpy().copy(), lab0.asnumpy().copy(), r.feat.asnumpy().copy())
        lg1, lab1 = r.step(1, data[1], 2)
        ref_cur = (lg1.asnumpy().copy(), lab1.asnumpy().copy(), r.feat.asnumpy().copy())
        k = model.key_forward(frames[0], want=("feat", "logits", "labels"))
        np.testing.assert_array_equal(k["logits"], ref_key[0])
        np.testing.assert_array_equal(k["labels"], ref_key[1].astype(np.uint8))
        np.testing.assert_array_equal(k["feat"], ref_key[2])
        c = model.cur_forward(frames[1], frames[0], want=("feat", "logits", "labels"))
        np.testing.assert_array_equal(c["logits"], ref_cur[0])
        np.testing.assert_array_equal(c["labels"], ref_cur[1].astype(np.uint8))
        np.testing.assert_array_equal(c["feat"], ref_cur[2])
    finally:
        tester.release_models()


def test_resident_input_reuse_is_invisible(demo_cfg):
    """Predictor.predict skips the PCIe copy of an input that is already in HBM (data_key = the array the previous call
    uploaded as data).  Results must be bit-identical to feeding fresh copies; arrays are immutable copies of their
    source (mx.nd.array copies like MXNet), so an edit of the source after the array was built must not leak in, and an
    edit through asnumpy() is refused."""
    from accel_amd import demo, mx
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 3)
    data = demo.build_batches(frames, demo_cfg)
    assert data[1][1] is data[0][0]          # the reuse precondition the demo loop creates (one array, two roles)
    try:
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        a = [r.step(i, data[i], 3)[0].asnumpy().copy() for i in range(3)]
        fresh = [[mx.nd.array(x.asnumpy().copy()) for x in row] for row in data]
        b = [r.step(i, fresh[i], 3)[0].asnumpy().copy() for i in range(3)]
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        src = data[2][0].asnumpy().copy()
        arr = mx.nd.array(src)
        src += 3.0                                       # the source changes after the array was built: no effect
        r.step(0, data[0], 3)
        r.step(1, data[1], 3)
        c = r.step(2, [arr, data[2][1], data[2][2]], 3)[0].asnumpy()
        np.testing.assert_array_equal(c, a[2])
        with pytest.raises(ValueError):
            arr.asnumpy()[...] += 1.0                    # handed out read-only
        r.step(0, data[0], 3)
        r.step(1, data[1], 3)
        d = r.step(2, [mx.nd.array(src), data[2][1], data[2][2]], 3)[0].asnumpy()      # new content, new array
        assert float(np.abs(d - a[2]).max()) > 1e-3
    finally:
        tester.release_models()


def test_pinned_prefetch_pipeline_is_invisible(demo_cfg):
    """Page-locked frames with the next frame's upload started beside the running forward (ClipRunner.prefetch ->
    accel_model_prefetch / accel_model_commit) give the same logits as the plain synchronous loop."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 5)
    try:
        plain = demo.build_batches(frames, demo_cfg)
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        ref = [r.step(i, plain[i], 3)[0].asnumpy().copy() for i in range(5)]
        pinned = demo.build_batches(frames, demo_cfg, pinned=True)
        assert pinned[0][0].pinned is not None
        got = []
        for i in range(5):
            lg, lab = r.step(i, pinned[i], 3)
            if i + 1 < 5:
                assert r.prefetch(pinned[i + 1])
            got.append((lg.asnumpy().copy(), lab.asnumpy().copy()))
        for (x, lab), y in zip(got, ref):
            np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(lab[0], np.argmax(x[0], axis=0))
        # a prefetched frame that is NOT the one fed next must not be used
        r.step(0, pinned[0], 3)
        r.prefetch(pinned[3])
        lg = r.step(1, pinned[1], 3)[0].asnumpy()
        np.testing.assert_array_equal(lg, ref[1])
    finally:
        tester.release_models()


def test_stale_feature_handle_is_never_read_silently(demo_cfg):
    """A feature handle is valid until the next forward that writes ITS buffer (non-key plans ping-pong between `feat`
    and `feat_b`, so the key feature survives one non-key frame and is overwritten by the second).  Reusing an older
    handle (DFF-style: the key feature for several non-key frames) must either still be valid, use the handle's host
    copy, or fail -- and two runners interleaving clips of the same size must not see each other's feature."""
    from accel_amd import demo, runtime
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    A = demo.build_batches(synth.make_clip(H, W, 3, seed=5), demo_cfg)
    B = demo.build_batches(synth.make_clip(H, W, 3, seed=6), demo_cfg)
    try:
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        ref = [r.step(i, A[i], 3)[0].asnumpy().copy() for i in range(3)]
        # (0) after ONE non-key frame the key feature is still in `feat` (the warp went to `feat_b`): reuse is legal and
        #     equals uploading a host copy of the key feature
        r.step(0, A[0], 3)
        key_feat = r.feat
        assert key_feat.device_ref[1] == "feat"
        r.step(1, A[1], 3)
        assert r.feat.device_ref[1] == "feat_b"
        r.feat = key_feat
        reuse = r.step(2, A[2], 3)[0].asnumpy().copy()
        # (1) stale handle without a host copy: refused
        r.step(0, A[0], 3)
        key_feat = r.feat
        r.step(1, A[1], 3)                      # feat -> feat_b
        r.step(2, A[2], 3)                      # feat_b -> feat: key_feat no longer describes the buffer
        assert r.feat.device_ref[1] == "feat"
        r.feat = key_feat
        with pytest.raises(runtime.AccelError, match="overwritten"):
            r.step(2, A[2], 3)
        with pytest.raises(runtime.AccelError, match="stale"):
            key_feat.asnumpy()
        # (2) the same pattern with a host copy taken in time: the copy is uploaded, result = warping the KEY feature
        r.step(0, A[0], 3)
        key_feat = r.feat
        key_host = key_feat.asnumpy().copy()
        r.step(1, A[1], 3)
        r.feat = key_feat                       # has a host copy now
        dff_style = r.step(2, A[2], 3)[0].asnumpy().copy()
        r.step(0, A[0], 3)
        from accel_amd import mx
        r.feat = mx.nd.array(key_host)
        expect = r.step(2, A[2], 3)[0].asnumpy()
        np.testing.assert_allclose(dff_style, expect, rtol=0, atol=1e-5 * max(1.0, float(np.abs(expect).max())))
        np.testing.assert_allclose(reuse, expect, rtol=0, atol=1e-5 * max(1.0, float(np.abs(expect).max())))
        # (3) two runners, same size and weights, clips interleaved frame by frame
        r2 = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        refB = [r2.step(i, B[i], 3)[0].asnumpy().copy() for i in range(3)]
        for i in range(3):
            a = r.step(i, A[i], 3)[0].asnumpy().copy()
            b = r2.step(i, B[i], 3)[0].asnumpy().copy()
            np.testing.assert_array_equal(a, ref[i])
            np.testing.assert_array_equal(b, refB[i])
    finally:
        tester.release_models()


# (inside `-m gpu` the same statement is made at the headline size: test_configs_gpu.py test_config4_batch8_1024x2048_equals_eight_single_clip_runs)
@pytest.mark.parametrize("version", [pytest.param("18", marks=pytest.mark.gpu_extra), "101"])
def test_batched_clips_match_single_clip_runs(demo_cfg, version):
    """Throughput mode: every call runs one frame of each of B independent clips (arrays with a leading batch of B).
    Image b of the batched run must reproduce the batch-1 run of clip b (same kernels, other tile choices: compared
    at 1e-4 of the logit range; labels identical outside the tie band) -- over a key frame and two non-key frames, so
    the per-image feature / featG hand-off is covered too."""
    from accel_amd import demo, mx
    from accel_amd.core import tester
    H, W, B, interval = 128, 256, 3, 3
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    clips = [synth.make_clip(H, W, interval, seed=77 + b) for b in range(B)]
    per_clip = [demo.build_batches(c, demo_cfg) for c in clips]
    try:
        single = []
        for b in range(B):
            r = demo.ClipRunner(version, demo_cfg, arg, aux, (H, W))
            single.append([r.step(t, per_clip[b][t], interval)[0].asnumpy().copy() for t in range(interval)])
        tester.release_models()
        rb = demo.ClipRunner(version, demo_cfg, arg, aux, (H, W), batch=B)
        for t in range(interval):
            arrays = [mx.nd.array(np.concatenate([per_clip[b][t][i].asnumpy() for b in range(B)], axis=0)) for i in range(2)]
            arrays.append(mx.nd.array(np.zeros((B, 2048, 1, 1), np.float32)))
            logits, labels = rb.step(t, arrays, interval)
            lg, lab = logits.asnumpy(), labels.asnumpy()
            assert lg.shape == (B, 19, H, W) and lab.shape == (B, H, W)
            for b in range(B):
                ref = single[b][t][0]
                tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
                assert float(np.abs(lg[b] - ref).max()) <= tol, (version, t, b, float(np.abs(lg[b] - ref).max()), tol)
                srt = np.sort(ref, axis=0)
                safe = (srt[-1] - srt[-2]) > 2 * tol
                np.testing.assert_array_equal(lab[b][safe], np.argmax(ref, axis=0)[safe])
    finally:
        tester.release_models()


@pytest.mark.parametrize("version,key_interval", [("18", 5), ("101", 3), ("50", 5), ("34", 3)])
def test_train_symbol_forward(demo_cfg, version, key_interval):
    """get_train_symbol, forward only (accel_18.py:31-119, accel_101.py:31-102): `data_ref` = key frame + intermediate
    frames, `data` = the labelled frame; all frame pairs through ONE FlowNet batch, the key feature warped once per pair,
    then the usual correction; `softmax_output` against the oracle's restatement (probabilities: absolute 1e-4)."""
    from accel_amd import mx, symbols
    from accel_amd.core import tester
    H, W = 128, 256
    demo_cfg.TRAIN.KEY_INTERVAL = key_interval
    n_ref = key_interval - 1
    inst = getattr(getattr(symbols, "accel_" + version), "accel_" + version)()
    sym = inst.get_train_symbol(demo_cfg)
    assert sym.list_outputs() == ["softmax_output", "data_ref", "eq_flag"]
    shapes = {"data": (1, 3, H, W), "data_ref": (n_ref, 3, H, W), "eq_flag": (1,), "label": (1, H, W)}
    inst.infer_shape(shapes)
    arg, aux = synth.make_params(inst.arg_shape_dict, inst.aux_shape_dict, data_names=tuple(shapes))
    frames = _oracle_frames(synth.make_clip(H, W, key_interval), demo_cfg)
    data, data_ref = frames[-1], np.concatenate(frames[:-1], axis=0)
    try:
        pred = tester.Predictor(sym, ["data", "data_ref", "eq_flag"], ["label"], context=[mx.gpu(0)],
                                provide_data=[[("data", shapes["data"]), ("data_ref", shapes["data_ref"]), ("eq_flag", (1,))]],
                                provide_label=[[("label", shapes["label"])]], arg_params=arg, aux_params=aux)
        batch = mx.io.DataBatch(data=[[mx.nd.array(data), mx.nd.array(data_ref), mx.nd.array(np.zeros((1,), np.float32))]], label=[],
                                provide_data=[[("data", data.shape), ("data_ref", data_ref.shape), ("eq_flag", (1,))]])
        out = pred.predict(batch)[0]
        prob = out["softmax_output"].asnumpy()
        assert out["data_ref"].shape == data_ref.shape
        lab = mx.nd.argmax(out["softmax_output"], axis=1).asnumpy()
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.train_forward(P, version, data, data_ref)
    assert prob.shape == (1, 19, H, W)
    assert float(np.abs(prob - ref["softmax_output"]).max()) <= 1e-4
    np.testing.assert_allclose(prob.sum(axis=1), 1.0, atol=1e-5)
    srt = np.sort(ref["softmax_output"], axis=1)
    safe = ((srt[:, -1] - srt[:, -2]) > 1e-3)[0]
    np.testing.assert_array_equal(lab[0][safe], np.argmax(ref["softmax_output"], axis=1)[0][safe])


def test_demo_end_to_end_on_a_cityscapes_layout(tmp_path, demo_cfg, capsys):
    """`python -m accel_amd.demo --data DIR --params A B --out DIR` end to end (dff_deeplab/demo.py:107-284) on a directory
    laid out like Cityscapes (leftImg8bit_sequence/val/<city>/<city>_<seq>_<frame>_leftImg8bit.png, 30-frame snippets with
    the label on frame 19; gtFine/val/<city>/..._gtFine_trainIds.png) and two MXNet-format checkpoints that are merged
    like demo.py:192-195.  The ground truth of every labelled frame is the oracle's own label map, so the printed mIoU
    must be 100 -- frame selection, city-keyed label lookup, checkpoint merge, inference and evaluator in one pass."""
    from PIL import Image
    from accel_amd import demo
    from accel_amd.core import tester
    from accel_amd.utils import load_model
    H, W, interval, num_ex = 128, 256, 3, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    flow_names = [k for k in arg if k.startswith(("flow_", "conv2", "conv3", "conv4", "conv5", "conv6", "Convolution", "deconv", "upsample_flow"))]
    load_model.save_checkpoint(str(tmp_path / "accel-18"), 0, {k: v for k, v in arg.items() if k not in flow_names}, aux)
    load_model.save_checkpoint(str(tmp_path / "flownet"), 0, {k: arg[k] for k in flow_names}, {})
    cities = ["frankfurt", "lindau"]
    P = dict(arg)
    P.update(aux)
    for ci, city in enumerate(cities[:num_ex]):
        seq = tmp_path / "data" / "leftImg8bit_sequence" / "val" / city
        gt = tmp_path / "data" / "gtFine" / "val" / city
        seq.mkdir(parents=True)
        gt.mkdir(parents=True)
        clip = synth.make_clip(H, W, 30, seed=900 + ci)
        for t, f in enumerate(clip):
            Image.fromarray(f[:, :, ::-1]).save(str(seq / ("%s_000000_%06d_leftImg8bit.png" % (city, t))))
        # the demo keeps frames 19-(interval-1) .. 19 of the snippet: key frame first, labelled frame last
        sel = clip[19 - (interval - 1):20]
        ref = G.run_clip(P, "18", _oracle_frames(sel, demo_cfg), interval)
        Image.fromarray(ref[-1][1][0].astype(np.uint8)).save(str(gt / ("%s_000000_000019_gtFine_trainIds.png" % city)))
    try:
        demo.main(["--version", "18", "--interval", str(interval), "--num_ex", str(num_ex), "--data", str(tmp_path / "data"),
                   "--params", str(tmp_path / "accel-18-0000.params"), str(tmp_path / "flownet-0000.params"),
                   "--out", str(tmp_path / "seg")])
    finally:
        tester.release_models()
    out = capsys.readouterr().out
    assert "===> final mIoU 100.000" in out, out[-1500:]
    assert out.count("(cum) mIoU") == num_ex and "frames/s" in out
    pngs = sorted(os.listdir(str(tmp_path / "seg")))
    assert len(pngs) == num_ex * interval and pngs[0].startswith("seg_frankfurt_000000_000017")
    seg = np.asarray(Image.open(str(tmp_path / "seg" / "seg_lindau_000000_000019_leftImg8bit.png")))
    gt = np.asarray(Image.open(str(tmp_path / "data" / "gtFine" / "val" / "lindau" / "lindau_000000_000019_gtFine_trainIds.png")))
    assert seg.shape == gt.shape and float((seg != gt).mean()) < 1e-3          # the written palette PNG holds the label ids
