"""Accuracy budget of the Winograd-on-bf16 launch geometries (41 / 42): max |logit error| against the CPU oracle at 1024x2048
(Accel-18, a key frame and two non-key frames) with the geometry withheld from different groups of layers.
    python scripts/debug/wb3_budget.py "" "res4" "res3,res4" ...      (each argument = one ACCEL_WB3_SKIP value)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from accel_amd import demo
from accel_amd.core import tester
from accel_amd.utils import synth
from oracle import graphs as G
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_configs_gpu import _oracle_frames
from accel_amd.config.config import config as cfg, reset_config, update_config

version = os.environ.get("VERSION", "18")
H, W, interval = 1024, 2048, 3
reset_config()
update_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
cfg.SCALES[0] = (H, W)
arg, aux = synth.model_params(version, H, W, cfg)
frames = synth.make_clip(H, W, 3)
P = dict(arg); P.update(aux)
ref = G.run_clip(P, version, _oracle_frames(frames, cfg), interval)
for skip in sys.argv[1:] or [""]:
    os.environ["ACCEL_WB3_SKIP"] = skip
    try:
        outs = demo.run_clip(version, cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    es = []
    for (lg, lab), (rlg, rlab) in zip(outs, ref):
        e = np.abs(np.asarray(lg) - np.asarray(rlg))
        at = np.unravel_index(int(e.argmax()), e.shape)
        es.append((float(e.max()), int((e > 1e-3).sum()), float(np.abs(rlg).max()), int(at[-2]), int(at[-1])))
    print("skip=%-24r fold=%s" % (skip, os.environ.get("ACCEL_FOLD_LINEAR", "1")), " ".join("e=%.3g (n>1e-3: %d, |logit| %.0f, at %d,%d)" % t for t in es), flush=True)
