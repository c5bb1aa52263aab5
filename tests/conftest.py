import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / on the GPU box)")


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
