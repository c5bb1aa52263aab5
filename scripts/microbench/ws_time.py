"""Per-phase timers of one block (debug build: hipcc ... -DWS_TIMING of the kernel source, linked into build/libaccel_wst.so; the kernel
writes clock64() differences into its output buffer).  Moved from the untracked scratch/ of round 2 so that the numbers DESIGN.md
quotes can be regenerated."""
import sys, numpy as np, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from accel_amd import runtime
runtime.LIB_PATH = os.environ.get("ACCEL_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build", "libaccel_wst.so")
ctx = runtime.Context(0)
for (C, K, H, W, N, res) in ((64, 256, 256, 512, 8, 0), (64, 256, 256, 512, 8, 1), (128, 512, 128, 256, 8, 1)):
    m = runtime.Model(ctx)
    rng = np.random.default_rng(0)
    m.set_param("w_weight", (rng.standard_normal((K, C, 1, 1)) * 0.05).astype(np.float32))
    sfx = ":%d" % N
    t = "option graph=0\narena bytes=256\npbuf name=x bytes=%d\npbuf name=r bytes=%d\npbuf name=y bytes=%d\n" % (N * H * W * C * 4, N * H * W * K * 4, N * H * W * K * 4)
    t += "conv name=c in=x:0:%d:%d:%d:%d%s out=y:0:%d:%d:%d:%d%s %sw=w_weight act=1 cin=%d cout=%d mode=conv tile=60 k=1,1 s=1,1 p=0,0 d=1,1\n" % (
        C, C, H, W, sfx, K, K, H, W, sfx, ("res=r:0:%d:%d:%d:%d%s " % (K, K, H, W, sfx)) if res else "", C, K)
    plan = m.add_plan("b", t)
    m.write("x", rng.standard_normal((N, H, W, C)).astype(np.float32))
    plan.finalize()
    for _ in range(5): plan.run()
    ctx.sync()
    print("%d->%d res=%d kernel us: %.1f" % (C, K, res, plan.profile(10)[0] * 1e3))
    y = m.read("y", (N * H * W * K,))[:32].reshape(4, 8)
    for w in range(4):
        n = max(y[w, 4], 1)
        print("  wave %d: tiles %d; per tile: issue %.0f, mfma %.0f, wait+barrier %.0f, epilogue %.0f" % (w, n, y[w, 0] / n, y[w, 1] / n, y[w, 2] / n, y[w, 3] / n))
    m.close()
