import sys, numpy as np, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from accel_amd import runtime
ctx = runtime.Context(0)
def run(mode, C, K, H, W, k, tile, N=1, extra=""):
    m = runtime.Model(ctx)
    rng = np.random.default_rng(0)
    wshape = (C, K, 4, 4) if mode == "deconv2x" else (K, C, k, k)
    m.set_param("w_weight", (rng.standard_normal(wshape) * (2.0 / (C * 4)) ** 0.5).astype(np.float32))
    Ho, Wo = (2 * H, 2 * W) if mode == "deconv2x" else (H, W)
    sfx = "" if N == 1 else ":%d" % N
    t = "option graph=0\narena bytes=256\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (N * H * W * C * 4, N * Ho * Wo * K * 4)
    ks = "" if mode == "deconv2x" else " k=%d,%d s=1,1 p=%d,%d d=1,1" % (k, k, k // 2, k // 2)
    t += "conv name=c in=x:0:%d:%d:%d:%d%s out=y:0:%d:%d:%d:%d%s w=w_weight act=0 cin=%d cout=%d mode=%s tile=%d%s %s\n" % (C, C, H, W, sfx, K, K, Ho, Wo, sfx, C, K, mode, tile, ks, extra)
    plan = m.add_plan("b", t)
    m.write("x", rng.standard_normal((N, H, W, C)).astype(np.float32))
    plan.finalize()
    plan.run(); ctx.sync()
    y = m.read("y", (N, Ho, Wo, K)).copy()
    info = plan.ops()[0]
    m.close()
    return y, info
for (mode, C, K, H, W, k) in (("deconv2x", 512, 2048, 32, 64, 4), ("conv", 2048, 1024, 64, 128, 1), ("deconv2x", 386, 64, 64, 128, 4), ("conv", 512, 512, 32, 64, 3)):
    ref, _ = run(mode, C, K, H, W, k, 32)
    for tile in (70, 71, 72, 73, 74, 75):
        for extra in ("", "split_target=1024"):
            try:
                y, info = run(mode, C, K, H, W, k, tile, extra=extra)
            except Exception as e:
                print(mode, C, K, tile, extra, "ERR", str(e)[:80]); continue
            err = np.abs(y - ref).max() / max(1.0, np.abs(ref).max())
            bad = np.argwhere(np.abs(y - ref).max(axis=-1) > 1e-4 * max(1.0, np.abs(ref).max()))
            print("%-8s %4d->%4d %3dx%3d tile %d %-10s ksplit %d: rel err %.2e%s" % (mode, C, K, H, W, tile, extra, info["ksplit"], err,
                  "" if err < 1e-4 else "  BAD pixels %d, first %s" % (len(bad), bad[:3].tolist())))
