"""conv_b3d.hip (launch geometries 82-85, 88, 89): the implicit GEMM of an f16-mode layer with both operands brought in by LDS-DMA.
(Its bf16x3 / fp16x2 forms for fp32 layers lost every A/B against conv_b3r.hip and were removed in round 5; half VIEWS are covered by
tests/test_f16_storage_gpu.py.)"""
import os

import numpy as np
import pytest

from oracle import ops as O

pytestmark = pytest.mark.gpu

def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


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
