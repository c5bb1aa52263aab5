"""Does the scale the first convolution of the key plan used (accel_plan_op_range) match the largest |pixel| of the frame it was run on,
whatever the frames before it looked like?  python scripts/debug/range_history.py [reps]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from accel_amd import demo, mx
from accel_amd.config.config import config, update_config
from accel_amd.core import tester
from accel_amd.utils import image, synth

H, W = 256, 512
update_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
config.SCALES[0] = (H, W)
arg, aux = synth.model_params("18", H, W, config)
A = synth.make_clip(H, W, 2, seed=5101)
pre = lambda fr, f: [(np.float32(f) * image.transform(im, config.network.PIXEL_MEANS)).astype(np.float32) for im in fr]
seqs = [pre(A, 1.0), pre([np.zeros_like(A[0])] * 2, 1.0), pre(A, 8.0), pre(A, 0.01), pre(A, 1.0), pre(A, 100.0), pre(A, 1e-3)]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = demo.ClipRunner("18", config, arg, aux, (H, W))
    zero = mx.nd.array(np.zeros((1, 2048, 1, 1), np.float32))
    for si, frames in enumerate(seqs):
        prev = None
        for idx, im in enumerate(frames):
            cur = mx.nd.array(im)
            prev = prev or cur
            lg, _ = r.step(idx, [cur, prev, zero], 2)
            fin = bool(np.isfinite(lg.asnumpy()).all())
            plan = (r.key_predictor if idx == 0 else r.cur_predictor).plan_for(H, W, 1)[0]
            rg = plan.ranges()
            name = "conv1" if idx == 0 else "18_conv0"
            s = rg[name][0]
            amax = float(np.abs(im).max())
            ok = 2.0 ** 13 <= s * amax < 2.0 ** 14
            if not (ok and fin):
                bad += 1
                print("rep %d seq %d frame %d: %s scale %g x max %g = %g (want [8192, 16384)), finite logits %s" % (rep, si, idx, name, s, amax, s * amax, fin))
    tester.release_models()
print("mismatches:", bad)
