"""The pooled form of the stem kernel (conv_stem_b3.hip <H2, POOL>; accel_hip.cpp fuse_stem_pool): the 7x7 / stride-2 stem of a network
and the 3x3 / stride-2 max pooling behind it in ONE kernel -- ResNet-101 `conv1` + `pool1` (pad 0, 'full' convention:
resnet_v1_101_flownet_deeplab.py:577-585) and the ResNet-18/34 branch's `conv0` + `pooling0` (pad 1, the next unit's BatchNorm + ReLU as the
pool's epilogue).  Bars: bit-identical to the two separate kernels (same conv values, a maximum is exact), and the operator bar
against the oracle; ragged sizes in both directions, several images, more tiles than persistent blocks."""
import numpy as np
import pytest

from accel_amd import runtime
from oracle import ops as O

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def pooled_size(n, pad, full):
    return (-(-(n + 2 * pad - 3) // 2) if full else (n + 2 * pad - 3) // 2) + 1


def run_pair(ctx, x, w, bn, pad, pool_bn, monkeypatch, fuse, tile=51):
    """plan text of the pair as the lowering writes it (accel_amd/lower.py lower_pool): conv + pool, both marked"""
    monkeypatch.setenv("ACCEL_STEM_POOL", "1" if fuse else "0")
    N, _, H, W = x.shape
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    full = pad == 0
    Hp, Wp = pooled_size(Ho, pad, full), pooled_size(Wo, pad, full)
    al = lambda b: (b + 255) // 256 * 256
    o_c = al(N * H * W * 4 * 4)
    o_p = o_c + al(N * Ho * Wo * 64 * 4)
    m = runtime.Model(ctx)
    try:
        m.set_param("w_weight", w)
        for k, v in zip(("gamma", "beta", "moving_mean", "moving_var"), bn):
            m.set_param("b_" + k, v)
        if pool_bn is not None:
            for k, v in zip(("gamma", "beta", "moving_mean", "moving_var"), pool_bn):
                m.set_param("q_" + k, v)
        sfx = ":%d" % N
        t = "option graph=0 tune=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (o_p + al(N * Hp * Wp * 64 * 4), x.nbytes, N * 64 * Hp * Wp * 4)
        t += "import_nchw src=x:0:3:3:%d:%d%s dst=A:0:3:4:%d:%d%s\n" % (H, W, sfx, H, W, sfx)
        t += "conv name=c in=A:0:3:4:%d:%d%s out=A:%d:64:64:%d:%d%s w=w_weight act=1 cin=3 cout=64 mode=conv tile=%d k=7,7 s=2,2 p=3,3 d=1,1 bn=b eps=1e-5 fixg=0 fuse_pool=1\n" % (
            H, W, sfx, o_c, Ho, Wo, sfx, tile)
        t += "pool name=p in=A:%d:64:64:%d:%d%s out=A:%d:64:64:%d:%d%s kind=max k=3,3 s=2,2 p=%d,%d fused=1" % (o_c, Ho, Wo, sfx, o_p, Hp, Wp, sfx, pad, pad)
        t += " act=1 bn=q eps=2e-5 fixg=0\n" if pool_bn is not None else " act=0\n"
        t += "export_nchw src=A:%d:64:64:%d:%d%s dst=y:0:64:64:%d:%d%s\n" % (o_p, Hp, Wp, sfx, Hp, Wp, sfx)
        plan = m.add_plan("p", t)
        m.write("x", x)
        plan.finalize()
        plan.run()
        times = plan.profile(1)
        y = m.read("y", (N, 64, Hp, Wp)).copy()
        kinds = [o["kind"] for o in plan.ops()]
        skipped = times[kinds.index("pool")] == 0.0
    finally:
        m.close()
    return y, skipped


def bn_params(seed):
    r = np.random.default_rng(seed)
    return ((0.5 + r.random(64)).astype(np.float32), rnd(seed + 1, 64, scale=0.3), rnd(seed + 2, 64, scale=0.2), (0.5 + r.random(64)).astype(np.float32))


@pytest.mark.parametrize("pad,with_bn", [(0, False), (1, True), (1, False)])
@pytest.mark.parametrize("N,H,W", [(1, 64, 96), (2, 50, 70), (1, 16, 260), (3, 37, 131), (2, 250, 518), (1, 128, 256)])
def test_stem_pool_pair_fused_vs_separate_vs_oracle(ctx, N, H, W, pad, with_bn, monkeypatch):
    x, w = rnd(80, N, 3, H, W, scale=50.0), rnd(81, 64, 3, 7, 7, scale=(2.0 / 147) ** 0.5 / 50.0)
    bn, pbn = bn_params(5), (bn_params(9) if with_bn else None)
    fused, skipped = run_pair(ctx, x, w, bn, pad, pbn, monkeypatch, True)
    apart, skipped0 = run_pair(ctx, x, w, bn, pad, pbn, monkeypatch, False)
    assert skipped and not skipped0                      # the fused plan launches nothing for the pooling op
    assert np.array_equal(fused, apart)
    ref = O.relu(O.batchnorm(O.conv2d(x, w, None, 2, 3, 1), *bn, 1e-5))
    ref = O.pool2d(ref, "max", 3, 2, pad, "full" if pad == 0 else "valid")
    if with_bn:
        ref = O.relu(O.batchnorm(ref, *pbn, 2e-5))
    assert float(np.abs(fused - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_stem_pool_is_not_fused_on_other_geometries(ctx, monkeypatch):
    """the fp32-MFMA stem (geometry 50) and the bf16x3 form have no pooled form: the pair runs as two kernels"""
    x, w = rnd(80, 1, 3, 64, 96, scale=50.0), rnd(81, 64, 3, 7, 7, scale=0.002)
    y50, skipped = run_pair(ctx, x, w, bn_params(5), 1, None, monkeypatch, True, tile=50)
    assert not skipped
    monkeypatch.setenv("ACCEL_SPLIT", "b3")
    y51, skipped = run_pair(ctx, x, w, bn_params(5), 1, None, monkeypatch, True, tile=51)
    assert not skipped
    assert float(np.abs(y50 - y51).max()) <= 1e-4 * max(1.0, float(np.abs(y50).max()))
