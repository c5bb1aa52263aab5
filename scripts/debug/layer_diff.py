"""Layer-by-layer comparison of two HIP bindings of the same clip (8 clips per call against 1 clip per call): which layer is the first
whose output differs by more than fp32 summation-order noise?  Both plans are bound with ACCEL_ARENA_NO_REUSE=1, so every
intermediate tensor is still in the arena after the run; image 0 of every op's output is read back from both.

    python scripts/debug/layer_diff.py [y x]      # image pixel whose neighbourhood is reported per layer (default 632 2039)
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))


def read_view(plan, v, image=0):
    from accel_amd import runtime
    b = v.buf
    assert b.space == "A" and b.esize == 4
    per = b.H * b.W * b.Cs * 4
    out = np.empty(per, np.uint8)
    runtime.check(runtime.lib().accel_plan_arena_read(plan.handle, ctypes.c_size_t(b.off + (v.img0 + image) * per), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(per), None))
    return out.view(np.float32).reshape(b.H, b.W, b.Cs)[..., v.coff:v.coff + v.C]


def collect(pred, H, W, nb):
    plan, lw = pred.plan_for(H, W, nb)
    out = []
    for kind, args in lw.ops:
        for key in ("out", "out2"):
            v = args.get(key)
            if v is None or not hasattr(v, "buf") or v.buf.space != "A" or v.buf.esize != 4:
                continue
            out.append((kind, args.get("name", kind) + ("" if key == "out" else ":2"), read_view(plan, v)))
    tiles = {o["name"]: (o["tile"], o["ksplit"]) for o in plan.ops() if o["kind"] == "conv"}
    return out, tiles


def main():
    from accel_amd import demo, mx
    from accel_amd.config.config import config, update_config
    from accel_amd.core import tester
    from accel_amd.utils import synth
    update_config(os.path.join(HERE, "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
    py, px = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (632, 2039)
    H, W, B, interval = 1024, 2048, 8, 2
    os.environ["ACCEL_ARENA_NO_REUSE"] = "1"
    config.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, config)
    clips = [synth.make_clip(H, W, 3)[:interval]] + [synth.make_clip(H, W, interval, seed=4100 + b) for b in range(1, B)]
    per_clip = [demo.build_batches(c, config) for c in clips]
    got = {}
    for nb in (B, 1):
        rb = demo.ClipRunner("18", config, arg, aux, (H, W), batch=nb)
        try:
            for t in range(interval):
                if nb == 1:
                    arrays = per_clip[0][t]
                else:
                    arrays = [mx.nd.array(np.concatenate([per_clip[b][t][i].asnumpy() for b in range(nb)], axis=0)) for i in range(2)]
                    arrays.append(mx.nd.array(np.zeros((nb, 2048, 1, 1), np.float32)))
                rb.step(t, arrays, interval)
                pred = rb.key_predictor if t == 0 else rb.cur_predictor
                got[(nb, t)] = collect(pred, H, W, nb)
        finally:
            tester.release_models()
    for t in range(interval):
        (a8, t8), (a1, t1) = got[(B, t)], got[(1, t)]
        d1 = {n: a for _, n, a in a1}
        print("==== frame %d (%s plan): layer, geometry/split-K at 8 | at 1, |x|max, max|d|, max|d|/|x|max, where, max|d| within 1 px of (%d, %d)" % (t, "key" if t == 0 else "cur", py, px))
        for kind, name, x8 in a8:
            x1 = d1.get(name)
            if x1 is None or x1.shape != x8.shape:
                print("%-40s only in one binding" % name)
                continue
            d = np.abs(x8 - x1).max(axis=2)
            y, x = np.unravel_index(int(np.argmax(d)), d.shape)
            s = H // x8.shape[0]
            fy, fx = py // s, px // s
            near = float(d[max(0, fy - 1):fy + 2, max(0, fx - 1):fx + 2].max())
            rng = float(np.abs(x1).max())
            base = name.split(":")[0]
            print("%-40s %-5s %8s | %-8s %9.3g %9.3g %9.3g  (%d, %d)/%dx%d  near %9.3g" % (
                name, kind, t8.get(base, ""), t1.get(base, ""), rng, float(d.max()), float(d.max()) / max(rng, 1e-30), y, x, x8.shape[0], x8.shape[1], near))


if __name__ == "__main__":
    main()
