"""Per-phase timers of one block (debug build: hipcc ... -DSTEM_TIMING of the kernel source, linked into build/libaccel_stemt.so; the kernel
writes clock64() differences into its output buffer).  Moved from the untracked scratch/ of round 2 so that the numbers DESIGN.md
quotes can be regenerated."""
import sys, numpy as np, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from accel_amd import runtime
runtime.LIB_PATH = os.environ.get("ACCEL_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build", "libaccel_stemt.so")
ctx = runtime.Context(0)
H, W, N = 1024, 2048, 8
m = runtime.Model(ctx)
rng = np.random.default_rng(0)
m.set_param("w_weight", (rng.standard_normal((64, 3, 7, 7)) * 0.05).astype(np.float32))
Ho, Wo = H // 2, W // 2
al = lambda b: (b + 255) // 256 * 256
o_y = al(N * H * W * 16)
t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=y bytes=%d\n" % (o_y + al(N * Ho * Wo * 256), N * 3 * H * W * 4, N * Ho * Wo * 256)
t += "prep_rgb src=x:0:3:4:%d:%d:%d dst=A:0:3:4:%d:%d:%d H=%d W=%d\n" % (H, W, N, H, W, N, H, W)
t += "conv name=c in=A:0:3:4:%d:%d:%d out=y:0:64:64:%d:%d:%d w=w_weight act=1 cin=3 cout=64 mode=conv tile=50 k=7,7 s=2,2 p=3,3 d=1,1\n" % (H, W, N, Ho, Wo, N)
plan = m.add_plan("b", t)
m.write("x", rng.standard_normal((N, 3, H, W)).astype(np.float32))
plan.finalize()
for _ in range(5): plan.run()
ctx.sync()
print("kernel us:", plan.profile(10)[1] * 1e3)
y = m.read("y", (N * Ho * Wo * 64,))[:32].reshape(4, 8)
for w in range(4):
    n = y[w, 4]
    print("wave %d: tiles %d; per tile: load_issue %.0f, mfma(2 passes) %.0f, passes+epilogues %.0f, window store+barrier %.0f" % (w, n, y[w, 0] / n, y[w, 1] / n, y[w, 2] / n, y[w, 3] / n))
