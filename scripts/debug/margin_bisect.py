"""Where does the headline binding's parity margin go?  (round-5 review, item 1)

Runs the clip of tests/test_configs_gpu.py::test_config4_batch8_* (8 clips per call, 1024x2048, key + non-key frame) under a
list of environment variants in ONE process and prints, per variant and frame, the largest |logit - oracle| of image 0, where
it sits and which class it is in.  The oracle evaluates clip 0 once.

    python scripts/debug/margin_bisect.py [variant ...]      # default: every variant of VARIANTS
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))

OFFSET_KEY = ["res5a_branch2b_offset", "res5b_branch2b_offset", "res5c_branch2b_offset"]
OFFSET_CUR = ["18_res5a_branch2b_offset", "18_res5b_branch2b_offset"]

VARIANTS = {
    "default": {},
    "no_halo": {"ACCEL_WITHHOLD": "halo"},
    "no_wino_split": {"ACCEL_WITHHOLD": "winograd_split"},
    "no_split": {"ACCEL_WITHHOLD": "split"},
    "bf16x3": {"ACCEL_SPLIT": "b3"},
    "key_off_9": {"ACCEL_FORCE_TILE": ",".join("%s=9" % n for n in OFFSET_KEY)},
    "cur_off_9": {"ACCEL_FORCE_TILE": ",".join("%s=9" % n for n in OFFSET_CUR)},
    "batch1": {"_batch": "1"},
    "no_wino": {"ACCEL_WITHHOLD": "winograd"},
}


def main():
    from accel_amd import demo, mx
    from accel_amd.config.config import config, update_config
    from accel_amd.core import tester
    from accel_amd.utils import image, synth
    from oracle import graphs as G
    update_config(os.path.join(HERE, "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
    names = sys.argv[1:] or list(VARIANTS)
    H, W, B, interval = 1024, 2048, 8, 2
    os.environ["ACCEL_ARENA_NO_REUSE"] = "1"
    config.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, config)
    clips = [synth.make_clip(H, W, 3)[:interval]] + [synth.make_clip(H, W, interval, seed=4100 + b) for b in range(1, B)]
    per_clip = [demo.build_batches(c, config) for c in clips]
    P = dict(arg)
    P.update(aux)
    t0 = time.time()
    ref = G.run_clip(P, "18", [image.transform(f, config.network.PIXEL_MEANS).astype(np.float32) for f in clips[0]], interval)
    print("oracle: %.1f s" % (time.time() - t0), flush=True)
    for name in names:
        env = dict(VARIANTS[name])
        nb = int(env.pop("_batch", B))
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        t0 = time.time()
        try:
            rb = demo.ClipRunner("18", config, arg, aux, (H, W), batch=nb)
            for t in range(interval):
                if nb == 1:
                    arrays = per_clip[0][t]
                else:
                    arrays = [mx.nd.array(np.concatenate([per_clip[b][t][i].asnumpy() for b in range(nb)], axis=0)) for i in range(2)]
                    arrays.append(mx.nd.array(np.zeros((nb, 2048, 1, 1), np.float32)))
                logits, labels = rb.step(t, arrays, interval)
                lg = logits.asnumpy()[0]
                rlg = ref[t][0][0]
                d = np.abs(lg - rlg)
                c, y, x = np.unravel_index(int(np.argmax(d)), d.shape)
                emap = d.max(axis=0)
                over = [(thr, int((emap > thr).sum())) for thr in (3e-4, 5e-4, 7e-4)]
                tiles = ""
                if t == interval - 1:
                    pred = rb.cur_predictor
                    ops = pred.plan_for(H, W, nb)[0].ops()
                    tiles = " ".join("%s=%s" % (o["name"], o["tile"]) for o in ops if o["kind"] == "conv" and "offset" in o["name"])
                    kops = rb.key_predictor.plan_for(H, W, nb)[0].ops()
                    tiles += " | " + " ".join("%s=%s" % (o["name"], o["tile"]) for o in kops if o["kind"] == "conv" and "offset" in o["name"])
                    if os.environ.get("BISECT_ALL_TILES"):
                        tiles += "\n    cur plan: " + " ".join("%s=%s/%s" % (o["name"], o["tile"], o["ksplit"]) for o in ops if o["kind"] == "conv")
                print("%-14s frame %d: e=%.3e at class %d pixel (%d, %d); pixels over %s; p99.99=%.3e  %s"
                      % (name, t, float(d.max()), c, y, x, over, float(np.quantile(emap[::4, ::4], 0.9999)), tiles), flush=True)
        finally:
            tester.release_models()
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        print("%-14s %.1f s" % (name, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
