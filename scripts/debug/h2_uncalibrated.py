import sys, numpy as np
sys.path.insert(0, '.')
from accel_amd import demo
from accel_amd.config.config import config as cfg
from accel_amd.utils import synth
H, W = 512, 1024
cfg.SCALES[0] = (H, W)
for version in ("18", "101", "50"):
    arg, aux = synth.model_params(version, H, W, cfg)
    data = demo.build_batches(synth.make_clip(H, W, 3), cfg)
    r = demo.ClipRunner(version, cfg, arg, aux, (H, W))
    for t in range(3):
        r.step(t, data[t], 5)
    for name, pred in (("key", r.key_predictor), ("cur", r.cur_predictor)):
        plan, lw = pred.plan_for(H, W, 1)
        rg = plan.ranges()
        un = [k for k, (s, c) in rg.items() if not c]
        print(version, name, len(rg), "h2 convs, uncalibrated:", un)
    r.close()
