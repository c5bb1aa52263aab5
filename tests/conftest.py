import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / on the GPU box)")
    config.addinivalue_line("markers", "gpu_extra: GPU robustness tests beyond the parity rows of SURVEY.md 8 (more bindings, more aspect ratios, more models of a "
                                       "property another parametrisation already covers): NOT selected by `-m gpu` -- the driver's GPU step has a time limit of 1200 s (round 6: `-m gpu` 420 tests in 849-924 s by box, "
                                       "`-m gpu_extra` 16 tests in 282 s) --, run them with `-m gpu_extra` (or `-m \"gpu or gpu_extra\"`)")


_TEST_TUNE = os.path.join(os.path.dirname(__file__), "golden", "gfx950_tests.tune")


def _use_the_suites_launch_geometry_table():
    """Reproducible summation order for the TEST shapes too.  The table shipped beside the library covers the BASELINE workloads; every other
    shape a test binds (128x256 ... 512x1024 clips, the operator tests) would be decided by timing on the box that runs the suite, so two
    boxes could round a flow field differently -- and which border pixel of a warp flips its last bit is exactly what the largest parity
    errors depend on.  tests/golden/gfx950_tests.tune holds the decisions of one full run of `-m gpu` (made with
    ACCEL_TUNE_CACHE=<file>: an explicit file receives every decision); a private copy of it is this session's user table, so the suite
    replays the same geometries on every box.  A test that sets ACCEL_TUNE_CACHE itself, or a caller who has, is left alone."""
    if os.environ.get("ACCEL_TUNE_CACHE") or not os.path.exists(_TEST_TUNE):
        return
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="accel_tune_")
    dst = os.path.join(d, "gfx950.tune")
    shutil.copyfile(_TEST_TUNE, dst)
    os.environ["ACCEL_TUNE_CACHE"] = dst


_use_the_suites_launch_geometry_table()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """how many launch geometries this session replayed from a table and how many it decided by timing (GPU sessions only)"""
    try:
        from accel_amd import runtime
        if runtime._lib is None:
            return
        replayed, timed, shipped = runtime.tune_stats()
        terminalreporter.write_line("launch geometries: %d replayed from a table, %d decided by timing in this session (shipped table: %d entries, "
                                    "suite table: %s)" % (replayed, timed, shipped, os.environ.get("ACCEL_TUNE_CACHE", "none")))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """`-m gpu` means the parity suite: tests that carry gpu_extra (a pytest.param mark or a decorator, beside the module's gpu mark) are left out
    unless the mark expression names gpu_extra itself."""
    expr = config.getoption("markexpr", "") or ""
    if "gpu_extra" in expr:
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_extra") is not None else keep).append(it)
    if drop and "gpu" in expr:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0; the product path has no CPU fallback, so a missing
    library or device is a hard failure of any gpu-marked test."""
    from accel_amd import runtime
    c = runtime.Context(0)
    yield c
    c.close()


@pytest.fixture
def demo_cfg():
    """The reference's dff_deeplab_vid_demo.yaml values that the path reads
    (SCALES, PIXEL_MEANS, NUM_CLASSES, NUM_ANCHORS), as committed fixture."""
    from accel_amd.config.config import config, reset_config, update_config
    reset_config()
    update_config(os.path.join(os.path.dirname(__file__), "golden", "dff_deeplab_vid_demo.yaml"))
    return config
