"""dcn_cols kernel variants on the res5 layers of the step: ACCEL_DCN_ONE_TAP=1 (one tap per thread), ACCEL_DCN_TAPS=3 / 9.
    python scripts/microbench/dcn_time.py"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(HERE, '..', '..'))
    import numpy as np
    from accel_amd import runtime
    ctx = runtime.Context(0)
    for (name, N, C, H, W, dg) in (("res5 512ch 64x128 x8 dg1", 8, 512, 64, 128, 1), ("r18 res5 512ch 32x64 x8 dg4", 8, 512, 32, 64, 4), ("res5 x1", 1, 512, 64, 128, 1)):
        m = runtime.Model(ctx)
        rng = np.random.default_rng(0)
        al = lambda b: (b + 255) // 256 * 256
        o_off = al(N * H * W * C * 4); o_col = o_off + al(N * H * W * 18 * dg * 4 + 64)
        tot = o_col + al(N * H * W * 9 * C * 4)
        ocs = (18 * dg + 3) // 4 * 4
        t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=o bytes=%d\n" % (tot, N * C * H * W * 4, N * 18 * dg * H * W * 4)
        t += "import_nchw src=x:0:%d:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d:%d\n" % (C, C, H, W, N, C, C, H, W, N)
        t += "import_nchw src=o:0:%d:%d:%d:%d:%d dst=A:%d:%d:%d:%d:%d:%d\n" % (18 * dg, 18 * dg, H, W, N, o_off, 18 * dg, ocs, H, W, N)
        t += "dcn_cols name=d in=A:0:%d:%d:%d:%d:%d off=A:%d:%d:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d:%d k=3,3 s=1,1 p=2,2 d=2,2 dg=%d bytes=%g\n" % (
            C, C, H, W, N, o_off, 18 * dg, ocs, H, W, N, o_col, 9 * C, 9 * C, H, W, N, dg, 2.0 * N * H * W * 9 * C * 4)
        plan = m.add_plan("b", t)
        m.write("x", rng.standard_normal((N, C, H, W)).astype(np.float32))
        m.write("o", (rng.standard_normal((N, 18 * dg, H, W)) * 1.2).astype(np.float32))
        plan.finalize()
        for _ in range(30): plan.run()
        ctx.sync()
        ms = plan.profile(20)[2]
        print("%-30s %7.1f us  %5.0f GB/s (column bytes written / time)" % (name, ms * 1e3, N * H * W * 9 * C * 4 / ms / 1e6), flush=True)
        m.close()
else:
    for env in ({"ACCEL_DCN_ONE_TAP": "1"}, {"ACCEL_DCN_TAPS": "3"}, {"ACCEL_DCN_TAPS": "9"}, {"ACCEL_DCN_TAPS": "n"}):
        print(env, flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "run"], env=dict(os.environ, **env))
