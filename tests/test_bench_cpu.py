"""bench.py's pricing of the convolution launches (no GPU): which family a launch belongs to, how many matrix products it EXECUTES
per algorithmic multiply-add, and the split of the roofline into the matrix regime and the HBM regime."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def op(tile, mode=0, flops=2e11, nbytes=5e8, narrow=0, kind="conv"):
    return {"kind": kind, "name": "x", "flops": flops, "bytes": nbytes, "tile": tile, "ksplit": 1, "narrow": narrow, "mode": mode}


def test_an_f16_layer_on_a_bf16x3_geometry_is_priced_at_one_product():
    """round 3 filed every launch on geometries 70-81 under the six-product family whatever the layer's mode and so reported
    0.69 of the fp16 peak for a kernel that executes 0.12 of it"""
    for tile in (70, 76, 79, 81, 82, 85):
        assert bench.conv_family(op(tile, mode=1), "f16") == "conv_f16_kernel"
        assert bench.conv_family(op(tile, mode=0), "f32") == "conv_igemm_b3_kernel"
    assert bench.PIPES["conv_f16_kernel"][0] == 1.0 and bench.PIPES["conv_igemm_b3_kernel"][0] == 6.0
    # 2e11 algorithmic flop in 1 ms = 200 TFLOP/s algorithmic
    r16 = bench.roofline_from_launches([(op(76, mode=1), 1.0, 1)], "f16")
    assert r16["kernel"] == "conv_f16_kernel" and r16["frac"] == pytest.approx(200.0 / 2500.0)
    assert r16["all_conv"]["frac"] == pytest.approx(0.08)
    r32 = bench.roofline_from_launches([(op(76, mode=0), 1.0, 1)], "f32")
    assert r32["kernel"] == "conv_igemm_b3_kernel" and r32["frac"] == pytest.approx(6 * 200.0 / 2500.0)
    # an op record without `mode` (older library) falls back to the plan's dtype
    legacy = {k: v for k, v in op(76).items() if k != "mode"}
    assert bench.conv_family(legacy, "f16") == "conv_f16_kernel" and bench.conv_family(legacy, "f32") == "conv_igemm_b3_kernel"
    # layers an f16 plan leaves in fp32 (RGB stems, ragged concats: mode 0) stay on the fp32 pipe
    assert bench.conv_family(op(3, mode=0), "f16") == "conv_igemm_f32_kernel"


def test_families_of_the_other_geometries():
    fam = lambda t, **k: bench.conv_family(op(t, **k), "f32")
    assert fam(40) == "conv_wino_f32_kernel" and fam(41) == fam(42) == fam(43) == "conv_wino_b3_kernel"
    assert fam(50) == "conv_stem_f32_kernel" and fam(60) == "conv1x1_ws_kernel" and fam(10) == "conv_igemm_f32_kernel"
    assert fam(51) == "conv_stem_b3_kernel" and bench.PIPES["conv_stem_b3_kernel"][0] == pytest.approx(6 * 176 / 147)
    assert fam(9, narrow=1) == "conv_narrow_kernel"
    assert bench.conv_family(op(3, mode=2), "bf16x3") == "conv_igemm_b3_kernel"        # bf16x3 plan option on a classic geometry id


def test_fp16x2_form_is_priced_at_three_products():
    """mode 3 = an fp32 layer in its fp16x2 form: geometries 76-81 and the Winograd geometries 41-43 execute THREE half products per
    multiply-add (the stem, 51, over its padded K); the geometries that have no such form (70-75, 82-87) still execute the six of the bf16 split"""
    fam = lambda t: bench.conv_family(op(t, mode=3), "f32")
    for t in (76, 77, 79, 80, 81):
        assert fam(t) == "conv_h2_kernel"
    assert fam(41) == fam(42) == fam(43) == "conv_wino_h2_kernel"
    assert fam(70) == fam(74) == fam(82) == fam(87) == "conv_igemm_b3_kernel"
    assert fam(51) == "conv_stem_h2_kernel" and bench.PIPES["conv_stem_h2_kernel"][0] == pytest.approx(3 * 176 / 147)
    assert fam(40) == "conv_wino_f32_kernel" and fam(10) == "conv_igemm_f32_kernel"
    assert bench.PIPES["conv_h2_kernel"][0] == 3.0 and bench.PIPES["conv_wino_h2_kernel"][0] == pytest.approx(3.0 / 2.25)
    r = bench.roofline_from_launches([(op(76, mode=3), 1.0, 1)], "f32")       # 200 TF algorithmic -> 600 executed
    assert r["kernel"] == "conv_h2_kernel" and r["frac"] == pytest.approx(3 * 200.0 / 2500.0)
    rb = bench.roofline_from_launches([(op(76, mode=0), 1.0, 1)], "f32")      # the same launch of a bf16x3 layer: 1200 executed
    assert rb["kernel"] == "conv_igemm_b3_kernel" and rb["frac"] == pytest.approx(6 * 200.0 / 2500.0)


def test_roofline_is_split_by_regime():
    """a launch that moves its algorithmic bytes at >= 3 TB/s is priced against HBM, the others against the matrix pipe"""
    deep = (op(76, flops=2e11, nbytes=5e8), 1.0, 4)            # 0.5 TB/s: matrix regime, 200 TF algorithmic
    short = (op(76, flops=1e10, nbytes=2e9), 0.5, 2)           # 4 TB/s: HBM regime
    pool = (op(-1, kind="pool", flops=0, nbytes=1e9), 0.3, 1)
    r = bench.roofline_from_launches([deep, short, pool], "f32")
    assert r["bound"] == "mfma" and r["launches_per_step"] == 4 and r["frac"] == pytest.approx(0.48)
    h = r["hbm_class"]
    assert h["bound"] == "hbm" and h["launches_per_step"] == 2 and h["achieved"] == pytest.approx(4000.0)
    assert h["frac"] == pytest.approx(0.5) and h["frac_of_achievable"] == pytest.approx(4000.0 / 6300.0, abs=1e-4)
    assert r["all_conv"]["matrix_regime_ms_per_step"] == pytest.approx(4.0) and r["all_conv"]["hbm_regime_ms_per_step"] == pytest.approx(1.0)
    assert r["all_conv"]["all_kernels_ms_per_step"] == pytest.approx(5.3)
    f = r["families"]["conv_igemm_b3_kernel"]
    assert f["matrix_regime"]["launches_per_step"] == 4 and f["hbm_regime"]["launches_per_step"] == 2
    assert r["all_conv"]["frac"] == pytest.approx(0.48)        # the HBM-regime launches no longer dilute the matrix fraction


def test_no_launch_is_priced_above_the_hbm_peak():
    """round 4 priced res3a_branch2a at 11.4 TB/s "algorithmic" (its byte count included the three quarters of the input a stride-2
    1x1 never reads) and let it inflate hbm_class.achieved: a launch above the 8 TB/s the part has is served by the caches and is
    reported in a class of its own, by name"""
    short = (op(76, flops=1e10, nbytes=2e9), 0.5, 2)           # 4 TB/s: HBM regime
    hot = (dict(op(76, flops=1e9, nbytes=1e9), name="res3a_branch2a"), 0.1, 1)      # 10 TB/s: cannot come from HBM
    r = bench.roofline_from_launches([short, hot, (op(76), 1.0, 1)], "f32")
    assert r["hbm_class"]["launches_per_step"] == 2 and r["hbm_class"]["achieved"] == pytest.approx(4000.0)
    assert r["cache_class"]["launches_per_step"] == 1 and r["cache_class"]["layers"] == ["res3a_branch2a (10000 GB/s)"]
    assert all(f["launches_per_step"] <= 3 for f in r["families"].values())
    assert r["hbm_class"]["achieved"] <= bench.HBM_PEAK_GBPS


def test_traffic_is_paired_with_the_algorithmic_bytes_of_the_same_launch_set():
    """roofline.traffic (PMC bytes per launch of every conv_b3r / conv_igemm_b3 launch) against the algorithmic bytes of THAT set --
    round 4's line put it beside the deep-K class's bytes, which read as 2.29x instead of 1.45x"""
    deep = (op(76, mode=3, flops=2e11, nbytes=3e8), 1.0, 1)       # matrix regime
    short = (op(77, mode=3, flops=1e10, nbytes=6e8), 0.15, 2)     # HBM regime, same rocprof kernel name
    wino = (op(43, mode=3, flops=1e11, nbytes=4e8), 0.5, 1)       # another kernel: not in the sampled set
    r = bench.roofline_from_launches([deep, short, wino], "f32")
    r = bench.pair_traffic(r, 750000000, "note", ["conv_igemm_b3_kernel", "conv_b3r_kernel"])
    assert r["traffic_algorithmic_bytes_per_launch"] == 500000000 and r["traffic_ratio"] == pytest.approx(1.5)
    assert r["algorithmic_bytes_per_launch"] == 300000000      # the deep-K class `achieved` is quoted on: a different set
    assert bench.pair_traffic(dict(r), None, None, [])["traffic"] is None
