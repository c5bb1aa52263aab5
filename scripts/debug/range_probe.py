"""Which range slots of a bound model's plans are non-finite / zero after a run (fp16x2 form, csrc/range.h)?
    python scripts/debug/range_probe.py 101 256 512"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from accel_amd import demo, mx
from accel_amd.config.config import config, update_config
from accel_amd.core import tester
from accel_amd.utils import image, synth

version, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
update_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
config.SCALES[0] = (H, W)
for v in sys.argv[4:] + [version]:
    arg, aux = synth.model_params(v, H, W, config)
    frames = [image.transform(f, config.network.PIXEL_MEANS).astype(np.float32) for f in synth.make_clip(H, W, 2, seed=5101)]
    r = demo.ClipRunner(v, config, arg, aux, (H, W))
    prev = None
    zero = mx.nd.array(np.zeros((1, 2048, 1, 1), np.float32))
    for idx, im in enumerate(frames):
        cur = mx.nd.array(im)
        prev = prev or cur
        lg, _ = r.step(idx, [cur, prev, zero], 2)
        print(v, "frame", idx, "finite logits:", bool(np.isfinite(lg.asnumpy()).all()))
        prev = cur
    for pred in (r.key_predictor, r.cur_predictor):
        plan = pred.plan_for(H, W, 1)[0]
        rg = plan.ranges()
        ops = {o["name"]: o for o in plan.ops()}
        bad = {k: (s, src, ops[k]["tile"]) for k, (s, src) in rg.items() if not (2.0 ** -99 < s < 2.0 ** 99)}
        print(v, plan.role, len(rg), "fp16x2 convs; measured by a pass:", sorted(k for k, (s, src) in rg.items() if src == 2),
              "; all-zero input:", sorted(k for k, (s, src) in rg.items() if src == 0), "; extreme scales:", bad)
    tester.release_models()
