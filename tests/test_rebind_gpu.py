"""Re-bind stress: an executor is deterministic per bind (dff_deeplab/core/module.py:1011-1044 binds once and every
forward of that binding computes the same function).  Every model is bound several times in one process -- fresh
context, streams, arena, weights each time -- at the bench size, and

  * every binding's frames must be bit-identical to the first binding's (launch geometries are replayed from the shipped
    table, so the summation order is fixed);
  * the captured hipGraph replay of a non-key frame must equal the SAME bound plan run op by op on one stream
    (accel_plan_run_serial), for the lowering with and without the linear folds.

This is the regression test of the round-2 "binding-dependent wrong frames" report.  What that was (DESIGN.md 7,
"two-stream hazard", logs in profiles/r03_twostream/): with TWO streams, whenever the two HIP streams of a binding landed on
different hardware queues, a kernel of the side stream could read 256-byte granules of an earlier same-stream kernel's output as
if they had not been written while a bandwidth-heavy kernel ran on the other queue.  No stand-alone reproducer was found, so the
two-stream lowering and its switch were REMOVED in round 4 (a switch that is wrong in 7 bindings of 8 has no place in the
product path); what stays is the contract: an executor is deterministic per bind."""
import os

import numpy as np
import pytest

from accel_amd.utils import synth

pytestmark = pytest.mark.gpu

H, W = 1024, 2048      # the bench size: where the round-2 failures were seen


def _bind_and_run(version, cfg, arg, aux, data, nframes=3):
    from accel_amd import demo
    r = demo.ClipRunner(version, cfg, arg, aux, (H, W))
    outs = [r.step(t, data[t], 5)[0].asnumpy() for t in range(nframes)]
    return r, outs


@pytest.mark.parametrize("version,binds", [("18", 2), pytest.param("18", 3, marks=pytest.mark.gpu_extra), pytest.param("34", 2, marks=pytest.mark.gpu_extra), pytest.param("50", 2, marks=pytest.mark.gpu_extra),
                                           ("101", 2), pytest.param("dff", 2, marks=pytest.mark.gpu_extra)])
def test_every_binding_computes_the_same_frames(demo_cfg, version, binds):
    from accel_amd import demo
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    data = demo.build_batches(synth.make_clip(H, W, 3), demo_cfg)
    ref = None
    for it in range(binds):
        r, outs = _bind_and_run(version, demo_cfg, arg, aux, data)
        try:
            if ref is None:
                ref = outs
            for t, (a, b) in enumerate(zip(outs, ref)):
                assert np.array_equal(a, b), "accel-%s binding %d frame %d differs from binding 0 by %g" % (
                    version, it, t, float(np.abs(a - b).max()))
        finally:
            r.close()


@pytest.mark.parametrize("fold", ["1", pytest.param("0", marks=pytest.mark.gpu_extra)])
def test_graph_replay_equals_the_serial_run_of_the_same_binding(demo_cfg, monkeypatch, fold):
    """The ping-pong variant-0 plan reads `feat` / `featG` and writes `feat_b` / `featG_b`: re-running it is idempotent,
    so the captured replay and the op-by-op run of one binding can be compared bit for bit."""
    from accel_amd import demo
    monkeypatch.setenv("ACCEL_FOLD_LINEAR", fold)
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    data = demo.build_batches(synth.make_clip(H, W, 2), demo_cfg)
    for it in range(2):
        r, outs = _bind_and_run("18", demo_cfg, arg, aux, data, nframes=2)
        try:
            plan, lw = r.cur_predictor.plan_for(H, W, 1)
            m = r.cur_predictor._model
            plan.run_serial()
            serial = m.read("logits", (1, 19, H, W))
            assert np.array_equal(outs[1], serial), "binding %d: replay and serial run differ by %g" % (
                it, float(np.abs(outs[1] - serial).max()))
        finally:
            r.close()
