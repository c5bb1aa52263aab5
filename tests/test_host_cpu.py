"""Host logic, no GPU: symbol surface and parameter tables, lowering/plan
invariants, config overlay, demo helpers, checkpoint IO, operator_py plugin
surface, C-ABI export check, and the rule that the product path never touches
the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")


def _sym(version):
    from accel_amd import symbols
    name = "accel_" + version
    return getattr(getattr(symbols, name), name)()


def _shapes(H, W, key):
    return {"data": (1, 3, H, W), "data_key": (1, 3, H, W), "feat_key": (1, 2048, 1, 1) if key else (1, 2048, H // 16, W // 16)}


# ---- symbol surface (accel_18.py:121-239 etc.) ------------------------------------------
@pytest.mark.parametrize("version", ["18", "34", "50", "101"])
def test_output_names_and_shapes(demo_cfg, version):
    inst = _sym(version)
    key = inst.get_key_test_symbol(demo_cfg)
    assert key.list_outputs() == ["data_key", "feat_key", "res5c_relu_output", "croped_score_output"]
    inst.infer_shape(_shapes(256, 512, True))
    assert inst.out_shape_dict["res5c_relu_output"] == (1, 2048, 16, 32)
    assert inst.out_shape_dict["croped_score_output"] == (1, 19, 256, 512)
    cur = inst.get_cur_test_symbol(demo_cfg)
    last = "croped_score_output" if version == "101" else "correction_output"
    assert cur.list_outputs() == ["data_key", "warping_feat_output", last]
    inst.infer_shape(_shapes(256, 512, False))
    assert inst.out_shape_dict["warping_feat_output"] == (1, 2048, 16, 32)
    assert inst.out_shape_dict[last] == (1, 19, 256, 512)
    assert {"data", "data_key", "feat_key"} <= set(cur.list_arguments())


def test_parameter_table_accel18(demo_cfg):
    """names / layouts a reference checkpoint must line up with"""
    inst = _sym("18")
    inst.get_cur_test_symbol(demo_cfg)
    inst.infer_shape(_shapes(128, 256, False))
    a, x = inst.arg_shape_dict, inst.aux_shape_dict
    assert a["corr_weight"] == (19, 38, 1, 1) and a["corr_bias"] == (19,)
    assert a["upsampling_weight"] == (19, 1, 32, 32) and a["18_upsampling_weight"] == (19, 1, 32, 32)
    assert a["18_feat_upsampling_weight"] == (512, 2048, 4, 4)
    assert a["fc6_weight"] == (1024, 2048, 1, 1) and a["18_score_weight"] == (19, 1024, 1, 1)
    assert a["18_res5a_branch2b_offset_weight"] == (72, 512, 3, 3) and a["18_res5a_branch2b_offset_bias"] == (72,)
    assert a["18_res5a_branch2b_weight"] == (512, 512, 3, 3) and "18_res5a_branch2b_bias" not in a
    assert a["18_conv0_weight"] == (64, 3, 7, 7) and a["18_stage1_unit1_sc_weight"] == (64, 64, 1, 1)
    assert a["18_stage3_unit1_conv1_weight"] == (256, 128, 3, 3)
    assert a["flow_conv1_weight"] == (64, 6, 7, 7) and a["deconv5_weight"] == (1024, 512, 4, 4)
    assert a["Convolution2_weight"] == (2, 1026, 3, 3) and a["upsample_flow6to5_weight"] == (2, 2, 4, 4)
    assert a["deconv4_weight"] == (1026, 256, 4, 4) and a["Convolution5_weight"] == (2, 194, 3, 3)
    assert not any(k.startswith("Convolution5_scale") for k in a)          # dead branch is pruned (ref F7)
    assert x["18_bn_data_moving_mean"] == (3,) and a["18_bn_data_gamma"] == (3,)
    assert "18_stage1_unit1_bn1_moving_var" in x and "18_bn5b_branch2b_moving_mean" in x
    assert "res5a_branch2a_weight" not in a                                 # no R101 trunk in the Accel-18 cur graph


def test_parameter_table_r101_and_variants(demo_cfg):
    inst = _sym("101")
    inst.get_key_test_symbol(demo_cfg)
    inst.infer_shape(_shapes(128, 256, True))
    a = inst.arg_shape_dict
    assert a["res5a_branch2b_offset_weight"] == (18, 512, 3, 3) and a["res5c_branch2b_weight"] == (512, 512, 3, 3)
    assert a["res3b3_branch2c_weight"] == (512, 128, 1, 1) and a["res4b22_branch2a_weight"] == (256, 1024, 1, 1)
    assert "res4b23_branch2a_weight" not in a and "res3b4_branch2a_weight" not in a
    assert a["res3a_branch1_weight"] == (512, 256, 1, 1) and a["conv1_weight"] == (64, 3, 7, 7)
    assert len([k for k in a if k.endswith("_weight")]) == 104 + 3 + 3    # 104 trunk convs (3 of them DCN) + 3 offset convs + fc6, score, upsampling
    inst.get_cur_test_symbol(demo_cfg)
    inst.infer_shape(_shapes(128, 256, False))
    assert inst.arg_shape_dict["corr_weight"] == (2048, 4096, 1, 1)
    i50 = _sym("50")
    i50.get_cur_test_symbol(demo_cfg)
    i50.infer_shape(_shapes(128, 256, False))
    a50 = i50.arg_shape_dict
    assert a50["curr_fc6_weight"] == (1024, 2048, 1, 1) and a50["curr_upsampling_weight"] == (19, 1, 32, 32)
    assert a50["50_res5a_branch2b_offset_weight"] == (72, 512, 3, 3) and a50["50_res4f_branch2c_weight"] == (1024, 256, 1, 1)
    i34 = _sym("34")
    i34.get_cur_test_symbol(demo_cfg)
    i34.infer_shape(_shapes(128, 256, False))
    assert "34_stage3_unit6_conv2_weight" in i34.arg_shape_dict and "34_res5c_branch2b_weight" in i34.arg_shape_dict
    assert "34_stage3_unit7_conv1_weight" not in i34.arg_shape_dict


def test_flownet_shape_trace_1024x2048(demo_cfg):
    """SURVEY.md appendix B"""
    from accel_amd.mx.symbol import infer_shapes
    inst = _sym("18")
    cur = inst.get_cur_test_symbol(demo_cfg)
    sh = infer_shapes(cur, _shapes(1024, 2048, False))
    byname = {n.name: sh[id(n)] for n in cur.topo() if id(n) in sh}
    exp = {"resize_data": (1, 6, 512, 1024), "flow_conv1": (1, 64, 256, 512), "conv2": (1, 128, 128, 256),
           "conv3_1": (1, 256, 64, 128), "conv4_1": (1, 512, 32, 64), "conv5_1": (1, 512, 16, 32),
           "conv6_1": (1, 1024, 8, 16), "deconv5": (1, 512, 18, 34), "crop_deconv5": (1, 512, 16, 32),
           "Concat2": (1, 1026, 16, 32), "Concat3": (1, 770, 32, 64), "Concat4": (1, 386, 64, 128),
           "deconv2": (1, 64, 130, 258), "Concat5": (1, 194, 128, 256), "resize_concat5": (1, 194, 64, 128),
           "Convolution5": (1, 2, 64, 128), "warping_feat": (1, 2048, 64, 128), "18_feat_upsampling": (1, 2048, 64, 128),
           "18_res5b_relu": (1, 512, 32, 64), "18_croped_score": (1, 19, 1024, 2048)}
    for k, v in exp.items():
        assert byname[k] == v, (k, byname[k], v)


# ---- lowering invariants -------------------------------------------------------------------
def _plan(version, key, H=128, W=256, **kw):
    from accel_amd import lower
    from accel_amd.config.config import config
    inst = _sym(version)
    sym = inst.get_key_test_symbol(config) if key else inst.get_cur_test_symbol(config)
    return lower.lower(sym, _shapes(H, W, key), **kw)


def test_algorithmic_flops_match_the_survey(demo_cfg):
    """BASELINE.md section 4 (GFLOP per frame at 1024x2048), within 0.5 %: the reference's layer list, i.e. the
    plan lowered WITHOUT the feat_upsampling*fc6 fold.  The fold removes exactly the 2048-channel intermediate:
    2*64*128*(4*512*2048 + 2048*1024 - 4*512*1024) flops (4 of the 16 taps reach each output pixel) on Accel-18/34;
    the warp/fc6 commutation removes the L head's fc6 (2*64*128*2048*1024) from every score-fusion non-key plan."""
    exp = {("18", True): 855.3, ("18", False): 381.2, ("34", False): 537.2, ("50", False): 679.3, ("101", False): 1076.8}
    saved = 2.0 * 64 * 128 * (4 * 512 * 2048 + 2048 * 1024 - 4 * 512 * 1024) / 1e9
    head = 2.0 * 64 * 128 * 2048 * 1024 / 1e9      # fc6 on the warped feature -> warp of the key frame's W*feat image
    for (v, key), gf in exp.items():
        _, lw = _plan(v, key, 1024, 2048, fold_linear=False)
        assert abs(lw.total_flops / 1e9 - gf) / gf < 5e-3, (v, key, lw.total_flops / 1e9, gf)
        assert not lw.derived and not lw.derived_bufs
        _, lwf = _plan(v, key, 1024, 2048)
        cut = (lw.total_flops - lwf.total_flops) / 1e9
        if key:
            assert cut == 0 and not lwf.derived and list(lwf.derived_bufs) == ["featG"]
        elif v in ("18", "34"):
            assert abs(cut - saved - head) < 1e-6 * saved and len(lwf.derived) == 1 and list(lwf.derived_bufs) == ["featG"]
        elif v == "50":
            assert abs(cut - head) < 1e-6 * head and not lwf.derived and list(lwf.derived_bufs) == ["featG"]
        else:      # Accel-101 fuses features (its fc6 runs on the `correction` output): the fusion over Concat(warped, current) keeps its
            # right half; the left half runs once per key frame (init:featC) and its image is warped (round 6, lower._plan_split_fusion)
            half = 2.0 * 64 * 128 * 2048 * 2048 / 1e9
            assert abs(cut - half) < 1e-6 * half and sorted(lwf.derived) == ["corr_weight[:,0:2048]", "corr_weight[:,2048:4096]"]
            assert list(lwf.derived_bufs) == ["featC"] and lwf.derived_bufs["featC"]["w"] == "corr_weight[:,0:2048]"


def test_fusion_over_a_concat_splits_into_its_two_halves_and_commutes_with_the_warp():
    """lower._plan_split_fusion (Accel-101's `correction`, accel_101.py:164-169) on the oracle's own operators:
    conv1x1(Concat(warp(F), X), W) + b  ==  warp(conv1x1(F, W[:, :c])) + conv1x1(X, W[:, c:]) + b, with the slices fold_params makes."""
    from accel_amd import lower
    from oracle import ops
    rng = np.random.RandomState(11)
    c, co, H, W = 8, 12, 6, 9
    F, X = rng.randn(1, c, H, W).astype(np.float32), rng.randn(1, c, H, W).astype(np.float32)
    flow = (rng.randn(1, 2, H, W) * 1.5).astype(np.float32)      # some taps leave the map
    w, b = (rng.randn(co, 2 * c, 1, 1) * 0.3).astype(np.float32), rng.randn(co).astype(np.float32)
    ref = ops.conv2d(np.concatenate([ops.flow_warp(F, flow), X], axis=1), w, b)
    d = lower.fold_params({"w[:,0:%d]" % c: ("cin_slice", "w", "0:%d" % c), "w[:,%d:%d]" % (c, 2 * c): ("cin_slice", "w", "%d:%d" % (c, 2 * c))}, {"w": w})
    wl, wr = d["w[:,0:%d]" % c], d["w[:,%d:%d]" % (c, 2 * c)]
    assert wl.shape == (co, c, 1, 1) and np.array_equal(np.concatenate([wl, wr], axis=1), w)
    got = ops.flow_warp(ops.conv2d(F, wl), flow) + ops.conv2d(X, wr, b)
    assert float(np.abs(got - ref).max()) <= 1e-5 * float(np.abs(ref).max())


def test_linear_fold_weight_is_the_composition():
    """lower.fold_params: deconv4x4/2(x, Wd) -> conv1x1(., Wf) + b  ==  deconv4x4/2(x, Wd*Wf) + b  (oracle ops)."""
    from accel_amd import lower
    from oracle import ops
    rng = np.random.RandomState(7)
    x = rng.randn(1, 12, 5, 6).astype(np.float32)
    wd = (rng.randn(12, 24, 4, 4) * 0.2).astype(np.float32)
    wf = (rng.randn(8, 24, 1, 1) * 0.2).astype(np.float32)
    b = rng.randn(8).astype(np.float32)
    two = ops.conv2d(ops.deconv2d(x, wd, None, stride=2, pad=1), wf, b)
    w = lower.fold_params({"a*b": ("deconv4x4s2*conv1x1", "a", "b")}, {"a": wd, "b": wf})["a*b"]
    assert w.shape == (12, 8, 4, 4) and w.dtype == np.float32
    one = ops.deconv2d(x, w, b, stride=2, pad=1)
    np.testing.assert_allclose(one, two, rtol=0, atol=2e-5 * float(np.abs(two).max()))


@pytest.mark.parametrize("version,key", [("18", True), ("18", False), ("34", False), ("50", False), ("101", False)])
def test_plan_is_well_formed(demo_cfg, version, key):
    text, lw = _plan(version, key)
    written, ranges = set(), []
    arena = int(re.search(r"^arena bytes=(\d+)", text, re.M).group(1))
    pbufs = dict(re.findall(r"^pbuf name=(\S+) bytes=(\d+)", text, re.M))
    for idx, line in enumerate(l for l in text.splitlines() if l and not l.startswith(("#", "arena", "pbuf", "option"))):
        kv = dict(t.split("=", 1) for t in line.split()[1:])
        for k, v in kv.items():
            m = re.match(r"^(\w+):(\d+):(\d+):(\d+):(\d+):(\d+)$", v)
            if not m:
                continue
            space, off, C, Cs, H, W = m.group(1), *map(int, m.groups()[1:])
            if space == "A":
                assert off % 16 == 0 and Cs % 4 == 0 and C <= Cs
                end = off + ((H * W - 1) * Cs + C) * 4
                assert end <= arena, line
                rng = (off, off + H * W * Cs * 4)
                if k in ("out", "out2", "dst"):
                    written.add(rng)
                elif k in ("in", "res", "left", "right", "feat", "flow", "off", "src"):
                    # every arena read must lie inside a region some earlier op wrote
                    assert any(w0 <= rng[0] and off + ((H * W - 1) * Cs + C) * 4 <= w1 for (w0, w1) in written), line
            else:
                assert space in pbufs, line
    # liveness plan: two buffers alive at the same op never overlap in the arena
    live = [b for b in lw.bufs if b.first is not None]
    for i, a in enumerate(live):
        for b in live[i + 1:]:
            if not (a.last < b.first or b.last < a.first):
                assert a.off + a.nbytes <= b.off or b.off + b.nbytes <= a.off, (a.id, b.id)
    kinds = [k for k, _ in lw.ops]
    assert kinds.count("score_tail") == 1
    if not key:
        n = 2     # beside the feature itself every model warps a linear image of it: featG = fc6_weight * feat, Accel-101 featC = corr_weight[:, :2048] * feat
        assert kinds.count("warp") == n and kinds.count("prep_flow") == 1 and kinds.count("copy") == n
    assert "Concat" not in text and all(k in ("prep_rgb", "prep_flow", "conv", "pool", "warp", "dcn_cols", "score_tail", "copy") for k in kinds)


def test_every_plan_is_lowered_for_one_stream(demo_cfg):
    """Plans carry no stream / wait annotations (the two-stream lowering of rounds 1-3 was removed: DESIGN.md 7), and the
    lowering no longer accepts the switch."""
    for version in ("18", "34", "50", "101"):
        for key in (True, False):
            text, lw = _plan(version, key)
            assert " stream=" not in text and " wait=" not in text
    with pytest.raises(TypeError):
        _plan("18", False, multi_stream=True)


@pytest.mark.parametrize("version", ["18", "50"])
def test_feature_pingpong_variants(demo_cfg, version):
    """Non-key graphs bind as two plans that hand the propagated feature back and forth between `feat`/`featG` and
    `feat_b`/`featG_b`: no copy-back of the warped feature, same work otherwise (accel_18.py:174-175 + demo.py:241-243)."""
    tb, base = _plan(version, False)
    (t0, v0), (t1, v1) = _plan(version, False, feat_slot=0), _plan(version, False, feat_slot=1)
    kinds = lambda lw: [k for k, _ in lw.ops]
    assert kinds(base).count("copy") == 2 and kinds(v0).count("copy") == 0 and kinds(v1).count("copy") == 0
    assert [k for k in kinds(base) if k != "copy"] == kinds(v0) == kinds(v1)
    assert v0.total_flops == v1.total_flops == base.total_flops
    pb = lambda t: set(re.findall(r"^pbuf name=(\S+)", t, re.M))
    assert pb(t0) == pb(t1) == pb(tb) | {"feat_b", "featG_b"}
    warps = lambda t: [dict(x.split("=", 1) for x in l.split()[1:]) for l in t.splitlines() if l.startswith("warp ")]
    w0, w1 = warps(t0), warps(t1)
    assert w0[0]["feat"].startswith("feat:") and w0[0]["out"].startswith("feat_b:")
    assert w1[0]["feat"].startswith("feat_b:") and w1[0]["out"].startswith("feat:")
    assert w0[1]["feat"].startswith("featG:") and w0[1]["out"].startswith("featG_b:")
    assert w1[1]["feat"].startswith("featG_b:") and w1[1]["out"].startswith("featG:")
    assert v0.outputs["warping_feat_output"].buf.space == "feat_b" and v1.outputs["warping_feat_output"].buf.space == "feat"
    # only `featG` is a derived buffer (rebuilt from `feat` after a host upload); the `_b` pair is written by plans only
    assert re.search(r"^pbuf name=featG bytes=\d+ from=feat$", t0, re.M) and "from=feat_b" not in t0 + t1
    # the key graph is not affected by the option
    assert _plan(version, True, feat_slot=0)[0] == _plan(version, True)[0]


def test_feature_pingpong_with_feature_fusion(demo_cfg):
    """Accel-101 (accel_101.py:161-166).  Lowered layer by layer (fold_linear=False) the warp writes straight into the Concat in front of
    the fusion convolution -- the warped feature is a slice of that buffer, so the plan keeps its one copy-back and the propagated feature
    stays in `feat`.  With the fusion split into its halves (the default, round 6: lower._plan_split_fusion) there is no Concat: the
    feature and the image featC = corr_weight[:, :2048] * feat ping-pong like feat / featG of the score-fusion models."""
    _, u0 = _plan("101", False, feat_slot=0, fold_linear=False)
    assert [k for k, _ in u0.ops].count("copy") == 1 and u0.outputs["warping_feat_output"].buf.space == "feat"
    (t0, v0), (t1, v1) = _plan("101", False, feat_slot=0), _plan("101", False, feat_slot=1)
    assert [k for k, _ in v0.ops].count("copy") == 0 and [k for k, _ in v0.ops] == [k for k, _ in v1.ops]
    assert v0.outputs["warping_feat_output"].buf.space == "feat_b" and v1.outputs["warping_feat_output"].buf.space == "feat"
    warps = lambda t: [dict(x.split("=", 1) for x in l.split()[1:]) for l in t.splitlines() if l.startswith("warp ")]
    w0, w1 = warps(t0), warps(t1)
    assert w0[1]["feat"].startswith("featC:") and w0[1]["out"].startswith("featC_b:")
    assert w1[1]["feat"].startswith("featC_b:") and w1[1]["out"].startswith("featC:")
    assert re.search(r"^pbuf name=featC bytes=\d+ from=feat$", t0, re.M) and "from=feat_b" not in t0 + t1
    # the fusion convolution reads the current frame's feature alone, the right half of the weight, and takes the warped image as residual
    conv = [dict(x.split("=", 1) for x in l.split()[1:]) for l in t0.splitlines() if l.startswith("conv ") and " name=correction " in l][0]
    assert conv["w"] == "corr_weight[:,2048:4096]" and conv["cin"] == "2048" and conv["res"].startswith("featC_b:") and conv["bias"] == "corr_bias"


def test_fusion_of_the_preactivation_units(demo_cfg):
    _, lw = _plan("18", False)
    convs = {a["name"]: a for k, a in lw.ops if k == "conv"}
    sc = convs["18_stage1_unit1_sc"]                      # shortcut conv + residual add, dual output for the next unit
    assert "res" in sc and sc["bn2"] == "18_stage1_unit2_bn1" and "out2" in sc
    assert convs["18_stage1_unit1_conv1"]["bn"] == "18_stage1_unit1_bn2" and convs["18_stage1_unit1_conv1"]["act"] == 1
    pools = [a for k, a in lw.ops if k == "pool" and a["kind"] == "max"]
    assert pools[0]["bn"] == "18_stage1_unit1_bn1" and pools[0]["act"] == 1
    rgb = [a for k, a in lw.ops if k == "prep_rgb"]
    assert rgb[0]["bn"] == "18_bn_data" and rgb[0]["fixg"] == 1
    assert convs["Convolution5"]["mul"] == 2.5
    assert convs["deconv5"]["mode"] == "deconv2x" and convs["deconv5"]["act"] == 2
    assert convs["18_res5a_branch2b"]["mode"] == "cols" and convs["18_res5b_branch2b"]["res"] is not None


def test_unsupported_graph_is_rejected_loudly(demo_cfg):
    from accel_amd import lower, mx
    d = mx.sym.Variable("data")
    c = mx.sym.Convolution(data=d, num_filter=8, kernel=(3, 3), num_group=2, name="g")
    with pytest.raises(NotImplementedError, match="grouped Convolution"):
        lower.lower(mx.sym.Group([c]), {"data": (1, 4, 128, 128)})


# ---- config / demo helpers ------------------------------------------------------------------
def test_update_config_rules(tmp_path, demo_cfg):
    from accel_amd.config.config import config, reset_config, update_config
    assert config.SCALES == [(1024, 2048)] and isinstance(config.network.PIXEL_MEANS, np.ndarray)
    assert config.network.NUM_ANCHORS == 9                      # sub-keys without a default are added freely
    bad = tmp_path / "bad.yaml"
    bad.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(ValueError, match="key must exist in config.py"):
        update_config(str(bad))
    train_like = tmp_path / "t.yaml"
    train_like.write_text("symbol: accel_18\nTRAIN:\n  KEY_INTERVAL: 5\n  arg_prefix: '18_'\nnetwork:\n  FIXED_PARAMS_SHARED: [conv1]\n")
    update_config(str(train_like))
    assert config.symbol == "accel_18" and config.TRAIN.KEY_INTERVAL == 5 and config.network.FIXED_PARAMS_SHARED == ["conv1"]
    reset_config()
    assert config.symbol == ""


def test_training_yaml_without_num_anchors_still_builds(demo_cfg):
    from accel_amd.config.config import config
    config.network.pop("NUM_ANCHORS", None)
    assert _sym("18").get_key_test_symbol(config).list_outputs()[-1] == "croped_score_output"


def test_frame_selection_and_miou_helpers():
    from accel_amd.demo import fast_hist, per_class_iu, select_frames
    names = ["f%03d" % i for i in range(90)]
    sel = select_frames(names, 3, 5, False)
    assert sel == names[15:20] + names[45:50] + names[75:80]         # offset = interval-1: labelled frame 19 is last
    sel = select_frames(names, 3, 5, True)
    assert sel == names[19:24] + names[48:53] + names[77:82]         # --avg: offset = i % interval
    pred = np.array([0, 1, 1, 2, 2, 2, 0])
    lab = np.array([0, 1, 2, 2, 2, 255, 1])                           # 255 = ignore
    h = fast_hist(pred, lab, 3)
    assert h.tolist() == [[1, 0, 0], [1, 1, 0], [0, 1, 2]]
    iu = per_class_iu(h)
    np.testing.assert_allclose(iu, [1 / 2, 1 / 3, 2 / 3])
    assert round(np.nanmean(per_class_iu(fast_hist(np.zeros(4, int), np.zeros(4, int), 3))) * 100, 2) == 100.0


def test_image_transform_and_resize(demo_cfg):
    from accel_amd.utils.image import resize, transform
    im = np.random.default_rng(0).integers(0, 255, (1024, 2048, 3)).astype(np.uint8)
    out, scale = resize(im, 1024, 2048, 0)
    assert scale == 1.0 and out is im                                  # Cityscapes frames pass through untouched
    t = transform(im[:4, :6], demo_cfg.network.PIXEL_MEANS)
    assert t.shape == (1, 3, 4, 6) and t.dtype == np.float64
    np.testing.assert_allclose(t[0, 0], im[:4, :6, 2] - 123.15)        # channel 0 = R = BGR[2] - means[2]
    np.testing.assert_allclose(t[0, 2], im[:4, :6, 0] - 103.06)
    small, s = resize(im[:512, :1024], 1024, 2048, 0)
    assert s == 2.0 and small.shape == (1024, 2048, 3)
    padded, _ = resize(im[:100, :130], 100, 130, stride=32)
    assert padded.shape == (128, 160, 3) and padded[100:].sum() == 0


# ---- checkpoint IO -----------------------------------------------------------------------------
@pytest.mark.parametrize("legacy", [True, False])
def test_params_roundtrip_and_prefix_rules(tmp_path, legacy):
    from accel_amd.utils import load_model as L
    d = {"arg:conv0_weight": np.random.rand(4, 3, 7, 7).astype(np.float32), "arg:fc6_bias_test": np.arange(5, dtype=np.float32),
         "aux:bn0_moving_mean": np.random.rand(4).astype(np.float32), "arg:18_conv1_weight": np.ones((2, 2), np.float32)}
    path = str(tmp_path / "m-0003.params")
    L.nd_save(path, d, legacy=legacy)
    back = L.nd_load(path)
    assert set(back) == set(d)
    for k in d:
        np.testing.assert_array_equal(back[k], d[k])
    arg, aux = L.load_param(str(tmp_path / "m"), 3, process=True, argprefix="18_")
    assert set(arg) == {"18_conv0_weight", "18_fc6_bias", "18_conv1_weight"} and set(aux) == {"18_bn0_moving_mean"}
    L.save_checkpoint(str(tmp_path / "s"), 0, {"w": d["arg:conv0_weight"]}, {"m": d["aux:bn0_moving_mean"]})
    a2, x2 = L.load_param(str(tmp_path / "s"), 0)
    np.testing.assert_array_equal(a2["w"], d["arg:conv0_weight"])
    np.testing.assert_array_equal(x2["m"], d["aux:bn0_moving_mean"])
    with open(path, "r+b") as f:
        f.write(b"\x00" * 8)
    with pytest.raises(ValueError, match="not an MXNet NDArray-list file"):
        L.nd_load(path)


# ---- operator_py plugin surface ---------------------------------------------------------------
def test_operator_registration_surface():
    import accel_amd.operator_py  # noqa: F401  (registers at import, like accel_18.py:11-15)
    from accel_amd import mx

    @mx.operator.register("scale_by")
    class ScaleProp(mx.operator.CustomOpProp):
        def __init__(self, factor="1"):
            super(ScaleProp, self).__init__(need_top_grad=False)
            self.factor = float(factor)

        def create_operator(self, ctx, shapes, dtypes):
            prop = self

            class Op(mx.operator.CustomOp):
                def forward(self, is_train, req, in_data, out_data, aux):
                    self.assign(out_data[0], req[0], in_data[0] * prop.factor)
            return Op()

    assert {"FlowWarp", "tile_as", "scale_by"} <= set(mx.operator.registered())
    s = mx.sym.Custom(mx.sym.Variable("x"), op_type="scale_by", factor=3)     # params arrive as strings
    assert s.attrs["params"] == {"factor": "3"} and s.infer_shape(x=(2, 3))[1] == [(2, 3)]
    op = s.attrs["prop"].create_operator(None, None, None)
    out = [np.zeros((2, 3))]
    op.forward(False, ["write"], [np.ones((2, 3))], out, [])
    np.testing.assert_array_equal(out[0], 3 * np.ones((2, 3)))
    op.forward(False, ["add"], [np.ones((2, 3))], out, [])
    np.testing.assert_array_equal(out[0], 6 * np.ones((2, 3)))
    with pytest.raises(KeyError, match="not registered"):
        mx.sym.Custom(mx.sym.Variable("x"), op_type="nope")
    t = mx.sym.Custom(data_shape=mx.sym.Variable("a"), data_content=mx.sym.Variable("b"), op_type="tile_as")
    assert t.infer_shape(a=(4, 3, 2, 2), b=(1, 5, 7))[1] == [(4, 5, 7)]
    top = t.attrs["prop"].create_operator(None, None, None)
    o = [np.zeros((4, 5, 7))]
    # the reference's interface (dff_deeplab/operator_py/tile_as.py:31-35): (data_content, data_shape) in THAT order -- positional
    # callers depend on it -- and the output is called data_tiled
    assert t.attrs["prop"].list_arguments() == ["data_content", "data_shape"] and t.attrs["prop"].list_outputs() == ["data_tiled"]
    top.forward(False, ["write"], [np.arange(35.0).reshape(1, 5, 7), np.zeros((4, 3, 2, 2))], o, [])
    assert (o[0][3] == np.arange(35.0).reshape(5, 7)).all()
    tp = mx.sym.Custom(mx.sym.Variable("b"), mx.sym.Variable("a"), op_type="tile_as")      # positional: content first
    assert tp.infer_shape(a=(4, 3, 2, 2), b=(1, 5, 7))[1] == [(4, 5, 7)]
    g = [np.ones((1, 5, 7)), np.ones((4, 3, 2, 2))]
    top.backward(["write", "write"], [np.ones((4, 5, 7))], [], [], g, [])
    assert not g[0].any() and not g[1].any()                                                # no gradient to either input


def test_flowwarp_alias_lowers_like_the_stock_pair(demo_cfg):
    import accel_amd.operator_py  # noqa: F401
    from accel_amd import lower, mx
    feat, flow = mx.sym.Variable("feat_key"), mx.sym.Variable("data")
    shapes = {"feat_key": (1, 2048, 8, 16), "data": (1, 3, 128, 256), "data_key": (1, 3, 128, 256)}
    pred = mx.sym.Convolution(data=flow, num_filter=2, kernel=(3, 3), stride=(16, 16), pad=(1, 1), name="p")
    a = mx.sym.BilinearSampler(data=feat, grid=mx.sym.GridGenerator(data=pred, transform_type="warp"), name="warping_feat")
    b = mx.sym.Custom(data=feat, flow=pred, op_type="FlowWarp", name="warping_feat")
    ta, _ = lower.lower(mx.sym.Group([a]), shapes)
    tb, _ = lower.lower(mx.sym.Group([b]), shapes)
    assert ta == tb and "\nwarp " in ta


# ---- C ABI ------------------------------------------------------------------------------------------
def test_cabi_exports_every_declared_symbol():
    from accel_amd import runtime
    hdr = open(os.path.join(ROOT, "include", "accel_hip.h")).read()
    declared = set(re.findall(r"\b(accel_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 28
    lib = ctypes.CDLL(runtime.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libaccel_hip.so does not export %s" % name
    runtime.lib()
    assert set(runtime.EXPORTS) <= declared


def test_product_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from accel_amd import runtime
    with pytest.raises(runtime.AccelError, match="no HIP device|HIP"):
        runtime.Context(0)
    assert runtime.lib().accel_sync(None) != 0 and b"NULL" in runtime.lib().accel_last_error()


def test_product_path_never_imports_the_oracle():
    """only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use oracle/"""
    pkg = os.path.join(ROOT, "accel_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|liboracle", src, re.M), os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle", bench)]
    assert len(uses) == 1 and bench.rfind("def cpu_baseline", 0, uses[0]) > bench.rfind("\ndef ", 0, bench.find("def cpu_baseline"))


# ---- Cityscapes IO / evaluator (lib/dataset/cityscape.py) -----------------------------------------
def test_cityscape_layout_palette_and_miou(tmp_path):
    from PIL import Image
    from accel_amd.dataset.cityscape import CityScape, confusion_matrix, getpallete
    root = tmp_path / "cityscapes"
    rng = np.random.default_rng(3)
    gts = {}
    for city, seq in (("frankfurt", "000000"), ("lindau", "000001")):
        (root / "leftImg8bit" / "val" / city).mkdir(parents=True)
        (root / "gtFine" / "val" / city).mkdir(parents=True)
        stem = "%s_%s_000019" % (city, seq)
        Image.fromarray(rng.integers(0, 255, (8, 16, 3)).astype(np.uint8)).save(str(root / "leftImg8bit" / "val" / city / (stem + "_leftImg8bit.png")))
        gt = rng.integers(0, 19, (8, 16)).astype(np.uint8)
        gt[0, :4] = 255                                        # ignore label
        Image.fromarray(gt).save(str(root / "gtFine" / "val" / city / (stem + "_gtFine_trainIds.png")))
        gts[stem] = gt
    db = CityScape("leftImg8bit_val", str(tmp_path), str(root), str(tmp_path / "out"))
    assert db.image_set_index == sorted(gts) and db.num_images == 2
    rec = db.load_segdb_from_index(db.image_set_index[0])
    assert (rec["height"], rec["width"]) == (8, 16) and rec["seg_cls_path"].endswith("_gtFine_trainIds.png")
    pal = getpallete(256)
    assert tuple(pal[:3]) == (128, 64, 128) and tuple(pal[18 * 3:18 * 3 + 3]) == (119, 11, 32) and pal[19 * 3:].sum() == 0
    perfect = [np.where(gts[i] == 255, 0, gts[i]) for i in db.image_set_index]
    info = db.evaluate_segmentations(perfect)
    present = np.unique(np.concatenate([g[g != 255] for g in gts.values()]))
    np.testing.assert_allclose(info["IU_array"][present], 1.0)
    png = Image.open(str(tmp_path / "out" / "results" / "frankfurt" / "frankfurt_000000_000019.png"))
    assert png.mode == "P" and np.array_equal(np.array(png), perfect[0])
    wrong = [np.full_like(p, 3) for p in perfect]
    info = db.evaluate_segmentations(wrong)
    assert info["IU_array"][3] > 0 and info["IU_array"][[c for c in present if c != 3]].sum() == 0
    cm = confusion_matrix(np.array([0, 1, 1, 2]), np.array([0, 1, 2, 2]), 3)
    assert cm.tolist() == [[1, 0, 0], [0, 1, 1], [0, 0, 1]]


def test_batched_lowering_scales_buffers_and_work(demo_cfg):
    """A leading batch of B frames (one per independent clip): every buffer reference carries `:B`, persistent buffers
    and the arena grow B-fold, per-op flops grow B-fold while weight bytes are counted once."""
    from accel_amd import lower
    from accel_amd.config.config import config
    inst = _sym("18")
    sym = inst.get_cur_test_symbol(config)
    sh1 = _shapes(128, 256, False)
    shB = {k: (3,) + tuple(v[1:]) for k, v in sh1.items()}
    t1, l1 = lower.lower(sym, sh1)
    tB, lB = lower.lower(sym, shB)
    assert abs(lB.total_flops - 3 * l1.total_flops) < 1e-6 * lB.total_flops
    assert lB.arena_bytes >= 3 * l1.arena_bytes - 3 * 256 * len(l1.bufs)
    for name, n1 in l1.pbufs.items():
        assert lB.pbufs[name] >= 3 * n1 - 256, name
    refs = re.findall(r"=([A-Za-z_]\w*:\d+:\d+:\d+:\d+:\d+(?::\d+)?)", tB)
    assert refs and all(r.endswith(":3") for r in refs), [r for r in refs if not r.endswith(":3")][:3]
    assert "feat_n=3" in tB and not re.search(r":\d+:\d+:\d+:\d+:\d+:\d+", t1)
    conv1 = [a for k, a in l1.ops if k == "conv"][3]
    convB = [a for k, a in lB.ops if k == "conv"][3]
    assert float(convB["bytes"]) < 3 * float(conv1["bytes"])      # the weights are read once per launch


def test_label_lookup_keeps_the_city():
    """Cityscapes val holds lindau_0000NN_000019 and munster_0000NN_000019 for the same NN: the ground-truth lookup of
    the demo must key on (city, sequence, frame), or every lindau frame is scored against a munster label."""
    from accel_amd import demo
    files = ["/d/gtFine/val/lindau/lindau_000003_000019_gtFine_trainIds.png",
             "/d/gtFine/val/munster/munster_000003_000019_gtFine_trainIds.png",
             "/d/gtFine/val/frankfurt/frankfurt_000001_000019_gtFine_trainIds.png"]
    table = {demo.label_key(f): f for f in files}
    assert len(table) == 3
    assert table[demo.label_key("/d/leftImg8bit_sequence/val/lindau/lindau_000003_000019_leftImg8bit.png")] == files[0]
    assert table[demo.label_key("/d/leftImg8bit_sequence/val/munster/munster_000003_000019_leftImg8bit.png")] == files[1]
    assert demo.label_key("/d/leftImg8bit_sequence/val/lindau/lindau_000003_000018_leftImg8bit.png") not in table
    assert demo.label_key("weird.png") is None


def test_nd_array_copies_and_is_immutable():
    """mx.nd.array copies its source (MXNet semantics) and hands host data out read-only; every array has its own uid,
    which is what input residency in HBM is keyed on"""
    from accel_amd import mx
    src = np.arange(12, dtype=np.float64).reshape(3, 4)
    a = mx.nd.array(src)
    src += 100.0
    assert a.asnumpy().dtype == np.float32 and float(a.asnumpy()[0, 1]) == 1.0
    with pytest.raises(ValueError):
        a.asnumpy()[0, 0] = 5.0
    b = mx.nd.array(src)
    assert a.uid != b.uid and mx.nd.array(a) is a


def test_params_token_is_content_keyed():
    """two predictors share a model only when their parameters are the same bytes (not the same dict object)"""
    from accel_amd.core.tester import params_token
    a = {"w": np.ones((2, 3), np.float32), "b": np.zeros(3, np.float32)}
    b = {"b": np.zeros(3, np.float32), "w": np.ones((2, 3), np.float32)}       # another dict, other order, same content
    assert params_token(a) == params_token(b)
    b["w"] = b["w"].copy()
    b["w"][1, 2] = 1.0000001
    assert params_token(a) != params_token(b)
    assert params_token(a, {}) == params_token(a) and params_token({"w": a["w"]}) != params_token({"v": a["w"]})
    assert params_token({"w": np.ones((3, 2), np.float32)}) != params_token({"w": np.ones((2, 3), np.float32)})


@pytest.mark.parametrize("version,key_interval", [("18", 5), ("34", 5), ("50", 5), ("101", 3)])
def test_train_symbol_structure_and_lowering(demo_cfg, version, key_interval):
    """get_train_symbol (accel_18.py:31-119, accel_101.py:31-102): argument / output names, shapes, and a lowering whose
    kernel list has ONE batched FlowNet pass over KEY_INTERVAL-1 frame pairs and as many chained warps."""
    from accel_amd import lower, symbols
    demo_cfg.TRAIN.KEY_INTERVAL = key_interval
    n = key_interval - 1
    inst = getattr(getattr(symbols, "accel_" + version), "accel_" + version)()
    sym = inst.get_train_symbol(demo_cfg)
    args = sym.list_arguments()
    assert sym.list_outputs() == ["softmax_output", "data_ref", "eq_flag"]
    for name in ("data", "data_ref", "eq_flag", "label", "corr_weight", "corr_bias", "fc6_weight", "flow_conv1_weight"):
        assert name in args, name
    assert "data_key" not in args and "feat_key" not in args
    H, W = 128, 256
    shapes = {"data": (1, 3, H, W), "data_ref": (n, 3, H, W), "eq_flag": (1,), "label": (1, H, W)}
    inst.infer_shape(shapes)
    assert inst.arg_shape_dict["corr_weight"] == ((2048, 4096, 1, 1) if version == "101" else (19, 38, 1, 1))
    _, outs, _ = sym.infer_shape(**shapes)
    assert outs[0] == (1, 19, H, W) and outs[1] == (n, 3, H, W)
    text, lw = lower.lower(sym, shapes)
    lines = text.splitlines()
    assert sum(l.startswith("warp ") for l in lines) == n
    flow1 = [l for l in lines if l.startswith("conv ") and "name=flow_conv1 " in l]
    assert len(flow1) == 1 and (":%d " % n) in flow1[0].replace("out=", " ").replace(" w=", ":X w=") or n == 1     # one batched launch
    assert sum(l.startswith("prep_flow") for l in lines) <= 2 and "softmax=1" in text
    assert lw.outputs["softmax_output"] == "logits" and lw.outputs["data_ref"] == "input:data_ref"
    # ResNet-101 runs once (18/34/50: on the key frame only; 101: key + current frame as one batch of 2)
    assert sum("name=res5c_branch2c " in l for l in lines) == 1


def test_init_weight_rules(demo_cfg):
    """accel_18.py:321-323, accel_50.py:310-318, accel_101.py:275-289"""
    from accel_amd.symbols.accel_18 import accel_18
    from accel_amd.symbols.accel_50 import accel_50
    from accel_amd.symbols.accel_101 import accel_101
    H, W = 128, 256
    for cls, ki in ((accel_18, 5), (accel_50, 5), (accel_101, 3)):
        demo_cfg.TRAIN.KEY_INTERVAL = ki
        inst = cls()
        inst.get_train_symbol(demo_cfg)
        inst.infer_shape({"data": (1, 3, H, W), "data_ref": (ki - 1, 3, H, W), "eq_flag": (1,), "label": (1, H, W)})
        arg = {"fc6_weight": np.full((1024, 2048, 1, 1), 2.0, np.float32), "fc6_bias": np.ones(1024, np.float32),
               "score_weight": np.ones((19, 1024, 1, 1), np.float32), "score_bias": np.ones(19, np.float32),
               "upsampling_weight": np.ones((19, 1, 32, 32), np.float32)}
        arg.update({"50_" + k: v * 3 for k, v in list(arg.items())})
        inst.init_weight(demo_cfg, arg, {}, rng=np.random.default_rng(0))
        assert arg["corr_bias"].shape == (2048 if cls is accel_101 else 19,) and not arg["corr_bias"].any()
        w = arg["corr_weight"]
        if cls is accel_101:
            # identity on the CURRENT frame's half of the stacked features, zero on the warped half
            assert w.shape == (2048, 4096, 1, 1) and not w[:, :2048].any()
            np.testing.assert_array_equal(w[:, 2048:, 0, 0], np.eye(2048, dtype=np.float32))
            assert arg["curr_fc6_weight"] is arg["fc6_weight"]
        else:
            assert w.shape == (19, 38, 1, 1) and 0.005 < float(w.std()) < 0.02
        if cls is accel_50:
            assert float(arg["curr_fc6_weight"][0, 0, 0, 0]) == 6.0 and arg["curr_upsampling_weight"] is arg["50_upsampling_weight"]
        if cls is accel_18:
            assert "curr_fc6_weight" not in arg


def test_shipped_tune_table_loads():
    """accel_amd/tune/gfx950.tune (launch geometries of the BASELINE workloads, regenerated by scripts/make_tune_table.sh)
    carries the version tag the library expects and is picked up from beside libaccel_hip.so."""
    import os
    import re
    from accel_amd import runtime
    path = os.path.join(os.path.dirname(runtime.LIB_PATH), "tune", "gfx950.tune")
    src = open(os.path.join(os.path.dirname(runtime.LIB_PATH), "csrc", "accel_hip.cpp")).read()
    tag = re.search(r'#define ACCEL_TUNE_VERSION "([^"]+)"', src).group(1)
    lines = open(path).read().splitlines()
    assert lines[0] == "# " + tag
    assert all(len(l.split()) == 19 for l in lines[1:]) and len(lines) > 100
    replayed, timed, shipped = runtime.tune_stats()
    assert shipped == len(lines) - 1 and timed == 0


def test_environment_switches_are_few_and_every_one_is_documented():
    """the product tree reads at most 20 ACCEL_* environment switches, each listed in INTEGRATION.md's table (kernel families are
    withheld through ONE list, ACCEL_WITHHOLD; bench.py takes its options as flags)"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for f in glob.glob(os.path.join(root, "accel_amd", "csrc", "*")) + glob.glob(os.path.join(root, "accel_amd", "**", "*.py"), recursive=True):
        if f.endswith((".cpp", ".hip", ".h", ".py")):
            src = open(f).read()
            found |= set(re.findall(r'getenv\("(ACCEL_[A-Z0-9_]+)"\)', src))
            found |= set(re.findall(r'environ(?:\.get|\.setdefault)?[\(\[]\s*"(ACCEL_[A-Z0-9_]+)"', src))
    assert 10 < len(found) <= 20, sorted(found)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [k for k in sorted(found) if "`" + k not in doc]
    assert not missing, "not in INTEGRATION.md's switch table: %s" % missing
    bench = open(os.path.join(root, "bench.py")).read()
    assert set(re.findall(r'"(ACCEL_BENCH_[A-Z_]+)"', bench)) | set(re.findall(r'(ACCEL_BENCH_[A-Z_]+)=', bench)) <= {"ACCEL_BENCH_SELF_LAUNCHED"}


_REF_CFGS = "/root/reference/experiments/dff_deeplab/cfgs"


@pytest.mark.skipif(not os.path.isdir(_REF_CFGS), reason="the reference checkout is only present in the build container")
def test_reference_experiment_configs_drop_in_unchanged():
    """experiments/dff_deeplab/cfgs/*.yaml load through config.update_config unchanged, name a symbol class this package
    has, and both test graphs of it build and lower (the harness's `eval(config.symbol + '.' + config.symbol)()`,
    demo.py:127-132).  Skipped where /root/reference does not exist (the GPU box)."""
    import glob
    from accel_amd import lower, symbols
    from accel_amd.config.config import config, reset_config, update_config
    files = sorted(glob.glob(os.path.join(_REF_CFGS, "*.yaml")))
    assert len(files) >= 5
    for f in files:
        reset_config()
        update_config(f)
        name = config.symbol or "accel_18"      # the demo yaml names no symbol: demo.py picks accel_<version> (demo.py:127)
        inst = getattr(getattr(symbols, name), name)()
        key, cur = inst.get_key_test_symbol(config), inst.get_cur_test_symbol(config)
        assert key.list_outputs()[-1] == "croped_score_output" and "warping_feat_output" in cur.list_outputs()
        H, W = 128, 256
        for sym, feat in ((key, (1, 2048, 1, 1)), (cur, (1, 2048, H // 16, W // 16))):
            shapes = {"data": (1, 3, H, W), "data_key": (1, 3, H, W), "feat_key": feat}
            shapes = {k: v for k, v in shapes.items() if k in sym.list_arguments()}
            text, lw = lower.lower(sym, shapes)
            assert lw.total_flops > 0 and "score_tail" in text
    reset_config()


def test_flow_field_layers_stay_off_the_winograd_bf16_geometries(demo_cfg):
    """Accuracy budget (accel_amd/lower.py): every convolution that feeds the flow input of the warp carries wb3=0
    (launch geometries 41 / 42 withheld), and no layer of the two ResNet branches does."""
    from accel_amd import lower
    from accel_amd.symbols import accel_18
    H, W = 256, 512
    sym = accel_18.accel_18().get_cur_test_symbol(demo_cfg)
    text, _ = lower.lower(sym, {"data": (1, 3, H, W), "data_key": (1, 3, H, W), "feat_key": (1, 2048, H // 16, W // 16)})
    tagged, free = set(), set()
    for line in text.split("\n"):
        if line.startswith("conv"):
            name = [t for t in line.split() if t.startswith("name=")][0][5:]
            (tagged if "wb3=0" in line.split() else free).add(name)
    assert {"conv3_1", "conv4_1", "conv5_1", "conv6_1", "flow_conv1", "Convolution5"} <= tagged
    # the other sampling positions of the path: the offset branches of the deformable layers (their producer alone, round 6)
    assert {n for n in tagged if n.startswith("18_")} == {"18_res5a_branch2b_offset", "18_res5b_branch2b_offset"}
    assert {"18_res5a_branch2a", "18_res5b_branch2a", "18_stage3_unit2_conv2"} <= free


@pytest.mark.parametrize("version", ["50", "101"])
def test_half_storage_set_of_the_f16_mode(demo_cfg, version):
    """f16-mode plans keep an arena buffer as half iff every writer is a half-capable convolution (or the deformable sampler writing
    its column buffer) and every reader a half-capable convolution reading it as input or residual (lower.Lowering.assign_storage,
    restated by oracle.graphs.STORE_F16).  Pinned here by NAME for the bottleneck trunks, written out independently of the rule:
    every branch convolution of res2..res5 stores half, except the res5 `branch2a` outputs (the offset convolution's narrow
    kernel and the deformable sampler read them as fp32) and the last block of a trunk whose output is a persistent buffer or
    feeds a non-convolution; the stems, pools, FlowNet's concat buffers, the heads' score maps and every persistent buffer stay fp32."""
    import re
    got = {}
    for key in (True, False):
        text, lw = _plan(version, key, H=256, W=512, conv_dtype="f16", store_f16=True, fold_linear=False)
        for kind, a in lw.ops:
            if kind == "conv":
                got[a["name"]] = (a["in"].buf.esize, a["out"].buf.esize, a["res"].buf.esize if "res" in a else None)
            elif kind == "dcn_cols":
                got[a["name"]] = (a["in"].buf.esize, a["out"].buf.esize, None)
            else:
                for k, v in a.items():
                    assert not hasattr(v, "buf") or v.buf.esize == 4, "only convolutions and column buffers touch half views: %s %s" % (kind, k)
        assert "A" not in [b.space for b in lw.half_bufs if b.space != "A"] and all(b.space == "A" for b in lw.half_bufs)
    trunks = [""] + (["50_"] if version == "50" else [])
    for name, (xin, out, res) in got.items():
        m = re.match(r"^(50_)?res(\d)(\w+?)_branch(1|2a|2b|2c)(_offset|_cols)?$", name)
        if not m or (m.group(1) or "") not in trunks:
            continue
        stage, blk, br, sfx = int(m.group(2)), m.group(3), m.group(4), m.group(5)
        if sfx == "_offset":
            assert (xin, out) == (4, 4), name                  # reads the fp32 branch2a output, writes the fp32 offsets
        elif sfx == "_cols":
            assert (xin, out) == (4, 2), name                  # the sampler reads fp32, the column buffer is half
        elif stage == 5 and br == "2a":
            assert out == 4, name
        elif stage == 5 and br == "2c" and blk == "c" and not (m.group(1) or ""):
            assert out == 4 and res == 2, name                 # res5c of the key trunk writes the persistent feature
        else:
            assert out == 2, name
        if br == "2c":
            assert res == 2, name                              # the shortcut (branch1 output or the previous block) is half
    for name in ("conv1", "flow_conv1", "conv2", "score", "Convolution1", "deconv5"):
        if name in got:
            assert got[name][1] == 4, name
    assert got["conv3"][1] == 2 and got["conv3_1"][:2] == (2, 4)      # FlowNet: conv3 -> conv3_1 is a plain chain, conv3_1 writes into a concat
    # with storage off nothing is half, and fp32 plans never are
    for kw in (dict(conv_dtype="f16", store_f16=False), dict(conv_dtype="f32", store_f16=True)):
        text, lw = _plan(version, False, H=256, W=512, **kw)
        assert ":h " not in text and not lw.half_bufs


def test_algorithmic_bytes_of_a_strided_1x1_count_only_the_sampled_pixels():
    """lower.py lower_anchor: res3a / res4a `branch1` and `branch2a` are 1x1 / stride 2 -- they read one pixel in four
    (resnet_v1_101_flownet_deeplab.py:646-660).  Round 4 counted the whole input and priced those launches above the HBM peak."""
    text, _ = _plan("18", True, 256, 512)
    kv = lambda l: dict(t.split("=", 1) for t in l.split()[1:] if "=" in t)
    ops = {kv(l)["name"]: kv(l) for l in text.split("\n") if l.startswith("conv ")}
    for name, cin, cout, ho, wo in (("res3a_branch2a", 256, 128, 32, 64), ("res3a_branch1", 256, 512, 32, 64), ("res4a_branch2a", 512, 256, 16, 32)):
        want = 4.0 * (ho * wo * cin + ho * wo * cout) + 4.0 * cin * cout
        assert float(ops[name]["bytes"]) == pytest.approx(want, rel=1e-5), name
    # a 3x3 / stride 2 layer reads everything
    b = ops["res3a_branch2b"] if ops["res3a_branch2b"]["s"] != "1,1" else ops["conv4"] if "conv4" in ops else None
    assert b is None or float(b["bytes"]) > 0
