"""BASELINE.json configs on hardware, one test per config ("configs not exercised"):

  config 1  Accel-18, 512x1024 two-frame pair, key-frame interval 1 (both frames key, the reference's own
            CPU-runnable case) and interval 2 (the warp / correction path) -- against the CPU oracle;
  config 2  Accel-18 1024x2048 against the CPU oracle at full size (key + two non-key frames, both lowerings);
  config 2/4  Accel-18 1024x2048, 8 clips per call (what bench.py times): the batched call against eight
            batch-1 runs of the same clips;
  config 3  Accel-101 1024x2048 against the CPU oracle at full size, and the size-independent properties of
            test_golden_gpu.py for the other model the BASELINE metric names;
  config 4  the RCCL gather of per-frame logits, on one GPU (world size 1): gathered bytes == logits buffer;
  config 5  Accel-50 with fp16-MFMA convolutions and half activation storage at 2048x4096, the first two frames of a key-frame
            interval 10 group (key frame = the ResNet-101 graph, non-key frame = FlowNet + warp + the Accel-50 branch): finite logits,
            non-degenerate label maps, and both frames against the mode's own specification (the oracle on half-rounded operands
            and half-rounded stored tensors, oracle.graphs ROUND_F16 + STORE_F16).
"""
import os
import socket

import numpy as np
import pytest

from accel_amd.utils import image, synth
from oracle import graphs as G

from parity_report import check_against_oracle, flip_windows, hip_border_points, logit_tolerance

pytestmark = pytest.mark.gpu


def _oracle_frames(frames_bgr, cfg):
    return [image.transform(f, cfg.network.PIXEL_MEANS).astype(np.float32) for f in frames_bgr]


_ORACLE_RUNS = {}


def _oracle_clip(P, version, frames_bgr, cfg, interval):
    """oracle.graphs.run_clip, remembered per (model, frames, interval) for the session: tests that look at the same clip from several
    sides (the three forced Winograd geometries; config 2 and image 0 of config 4) share ONE oracle evaluation -- the suite has to fit
    the driver's time limit (round 4: 1081 of 1200 s).  A longer run of the same clip serves a shorter one (frame t depends on frames <= t)."""
    import hashlib
    h = hashlib.sha1()
    for f in frames_bgr:
        h.update(np.ascontiguousarray(f).tobytes())
    key = (str(version), frames_bgr[0].shape, interval)
    for (k, n, digests), ref in _ORACLE_RUNS.items():
        if k == key and n >= len(frames_bgr) and digests[:len(frames_bgr)] == tuple(hashlib.sha1(np.ascontiguousarray(f).tobytes()).hexdigest() for f in frames_bgr):
            out = G.ClipResult(ref[:len(frames_bgr)])
            out.critical = ref.critical[:len(frames_bgr)]
            out.warp_border = ref.warp_border[:len(frames_bgr)]
            return out
    ref = G.run_clip(P, str(version), _oracle_frames(frames_bgr, cfg), interval)
    _ORACLE_RUNS[(key, len(frames_bgr), tuple(hashlib.sha1(np.ascontiguousarray(f).tobytes()).hexdigest() for f in frames_bgr))] = ref
    return ref


@pytest.mark.parametrize("interval", [1, 2])
def test_config1_accel18_512x1024_pair(demo_cfg, interval):
    """dff_deeplab demo on a 2-frame 512x1024 pair.  interval 1: `idx % 1 == 0` makes both frames key frames
    (demo.py:235), the cur graph is not touched; interval 2: frame 1 goes through FlowNet, warp, the R18 branch and
    the score fusion."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W = 512, 1024
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip("18", demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = _oracle_clip(P, "18", frames, demo_cfg, interval)
    check_against_oracle(outs, ref, "config1 accel-18 512x1024 kf=%d" % interval, margin_bar=0.5)


@pytest.mark.parametrize("geometry", [pytest.param(41, marks=pytest.mark.gpu_extra), 42, 43])
def test_winograd_bf16_geometry_on_every_eligible_layer_vs_oracle(demo_cfg, monkeypatch, tmp_path, geometry):
    """The launch-geometry table decides per layer; this test does not depend on what it decided: EVERY layer that can take
    the Winograd-on-bf16 geometry 41 / 42 / 43 (every 3x3 / stride 1 layer of the two ResNet branches with channels in
    multiples of 16 -- the FlowNet layers are withheld by the lowering) is forced onto it (ACCEL_WB3_FORCE: applied
    to the plans directly, the launch-geometry table is neither read nor written for those layers), and a key + a non-key frame of Accel-18 at 512x1024 are compared with
    the oracle at the whole-graph tolerance."""
    from accel_amd import demo, runtime
    from accel_amd.core import tester
    monkeypatch.setenv("ACCEL_WB3_FORCE", str(geometry))
    H, W, interval = 512, 1024, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    data = demo.build_batches(frames, demo_cfg)
    try:
        lib = runtime.lib()
        import ctypes
        before = [ctypes.c_int() for _ in range(3)]
        lib.accel_tune_stats(*[ctypes.byref(v) for v in before])
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        outs = []
        for idx, arrays in enumerate(data):
            logits, labels = r.step(idx, arrays, interval)
            outs.append((logits.asnumpy(), np.uint8(np.squeeze(labels.asnumpy()))))
        forced = 0
        for pred in (r.key_predictor, r.cur_predictor):
            forced += sum(1 for o in pred.plan_for(H, W, 1)[0].ops() if o["kind"] == "conv" and o["tile"] == geometry)
        assert forced >= 8, "the forced geometry must have been applied to the 3x3 layers (%d launches)" % forced
        # the forced geometry lives in the plans only: nothing was timed for those layers and the process-wide table is untouched,
        # so what later tests of this session replay does not depend on this one having run
        after = [ctypes.c_int() for _ in range(3)]
        lib.accel_tune_stats(*[ctypes.byref(v) for v in after])
        assert after[2].value == before[2].value
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = _oracle_clip(P, "18", frames, demo_cfg, interval)
    check_against_oracle(outs, ref, "accel-18 512x1024 every eligible layer on geometry %d" % geometry)


def test_config4_batch8_1024x2048_equals_eight_single_clip_runs(demo_cfg, monkeypatch):
    """What bench.py times: one call = one frame of each of 8 independent clips at 1024x2048.  Image b of the batched
    call must reproduce the batch-1 run of clip b over a key and a non-key frame (tile choices differ between the two
    binds, so sums may differ in the last bits: 1e-4 of the logit range; labels identical outside the tie band); image 0 of the
    batched call is also checked against the oracle's run of clip 0."""
    from accel_amd import demo, mx
    from accel_amd.core import tester
    H, W, B, interval = 1024, 2048, 8, 2
    # private space for every intermediate buffer: the offset maps of the deformable layers survive the run, and a
    # deviation between the two evaluations is tolerated only at a border discontinuity those offsets show
    monkeypatch.setenv("ACCEL_ARENA_NO_REUSE", "1")
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    # clip 0 is the clip of test_config2_* (its first two frames: key + non-key under both schedules), so that the oracle's evaluation of it is shared
    clips = [synth.make_clip(H, W, 3)[:interval]] + [synth.make_clip(H, W, interval, seed=4100 + b) for b in range(1, B)]
    per_clip = [demo.build_batches(c, demo_cfg) for c in clips]
    try:
        single = [[None] * interval for _ in range(B)]
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        checked = (0, 3, 7)      # images of the batched call that are compared with a single-clip run of their own (first, middle, last)
        for b in checked:
            for t in range(interval):
                lg, lab = r.step(t, per_clip[b][t], interval)
                # keep the sub-sampled logits and the full label map of every single run (8 x 160 MB otherwise)
                single[b][t] = (lg.asnumpy()[0][:, ::2, ::2].copy(), np.uint8(lab.asnumpy()[0]))
        tester.release_models()
        rb = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W), batch=B)
        carried, image0 = [], []
        for t in range(interval):
            arrays = [mx.nd.array(np.concatenate([per_clip[b][t][i].asnumpy() for b in range(B)], axis=0)) for i in range(2)]
            arrays.append(mx.nd.array(np.zeros((B, 2048, 1, 1), np.float32)))
            logits, labels = rb.step(t, arrays, interval)
            lg, lab = logits.asnumpy(), labels.asnumpy()
            assert lg.shape == (B, 19, H, W) and lab.shape == (B, H, W)
            image0.append((lg[0:1].copy(), lab[0:1].copy()))
            pred = rb.key_predictor if t % interval == 0 else rb.cur_predictor
            pts = hip_border_points(*pred.plan_for(H, W, B, slot=0))
            carried = pts if t % interval == 0 else carried + pts
            for b in checked:
                ref, rlab = single[b][t]
                tol = logit_tolerance(ref)
                emap = np.abs(lg[b][:, ::2, ::2] - ref).max(axis=0)
                # the two evaluations (other launch geometries, other summation order) agree to the logit tolerance, except
                # inside footprints of a border discontinuity that this run's own offsets show (parity_report.flip_windows)
                crit = [(name, n, y / 2, x / 2) for name, n, y, x in carried if n == b]
                flips, centres = flip_windows(emap, tol, win=32, critical=crit, radius=64)
                if centres:
                    print("batch-8 vs single, frame %d clip %d: verified discontinuity footprint(s) at %s" % (t, b, centres))
                assert float((np.uint8(lab[b]) != rlab).mean()) < 2e-3, (t, b)
    finally:
        tester.release_models()
    # ... and the batched call against the ORACLE directly: image 0 of the 8-clip call (the launch geometries bench.py replays:
    # M = 8x the rows, other table entries than the one-clip bind) must meet the same bar as a one-clip run does
    P = dict(arg)
    P.update(aux)
    ref = _oracle_clip(P, "18", clips[0], demo_cfg, 3)      # (interval 3 = config 2's schedule: frames 0, 1 are key, non-key as here)
    check_against_oracle(image0, ref, "config4 accel-18 1024x2048, image 0 of the 8-clip call", margin_bar=0.5)


def test_config2_accel18_1024x2048_vs_oracle(demo_cfg, monkeypatch):
    """BASELINE config 2 at the size the metric is quoted on (the reference validates at 1024x2048 only, README.md:60-71):
    a key frame and two non-key frames of Accel-18 against the CPU oracle, with the shipped launch-geometry table the
    bench replays, for the default lowering and for the reference's layer list run one to one (ACCEL_FOLD_LINEAR=0).
    Tolerance and the verified-discontinuity rule: tests/parity_report.py.  (The oracle needs ~50 s for the three frames
    on 32 host cores.)"""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 1024, 2048, 3
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 3)
    P = dict(arg)
    P.update(aux)
    ref = _oracle_clip(P, "18", frames, demo_cfg, interval)
    for mode in ("1", "0"):
        monkeypatch.setenv("ACCEL_FOLD_LINEAR", mode)
        try:
            outs = demo.run_clip("18", demo_cfg, arg, aux, frames, interval)
        finally:
            tester.release_models()
        check_against_oracle(outs, ref, "config2 accel-18 1024x2048 fold=%s" % mode, margin_bar=0.5)


def test_config3_accel101_1024x2048_vs_oracle(demo_cfg):
    """BASELINE config 3 at full size: a key and a non-key frame of Accel-101 (ResNet-101 on both branches, feature
    fusion 4096 -> 2048) against the CPU oracle."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 1024, 2048, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("101", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip("101", demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, "101", _oracle_frames(frames, demo_cfg), interval)
    check_against_oracle(outs, ref, "config3 accel-101 1024x2048", margin_bar=0.5)


@pytest.mark.parametrize("version", [pytest.param("34", marks=pytest.mark.gpu_extra), "50"])
# (Accel-34 inside `-m gpu`: against the oracle at 128x256 and 208x176, test_graph_gpu.py)
def test_accel34_accel50_1024x2048_vs_oracle(demo_cfg, version):
    """The other two models at the size the reference validates at: a key and a non-key frame against the CPU oracle."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 1024, 2048, 2
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params(version, H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    try:
        outs = demo.run_clip(version, demo_cfg, arg, aux, frames, interval)
    finally:
        tester.release_models()
    P = dict(arg)
    P.update(aux)
    ref = G.run_clip(P, version, _oracle_frames(frames, demo_cfg), interval)
    check_against_oracle(outs, ref, "accel-%s 1024x2048" % version, margin_bar=0.5)


def test_config3_accel101_full_size_properties_1024x2048(demo_cfg):
    """Accel-101 at the BASELINE size (the oracle takes minutes there): determinism, fused argmax == argmax of the
    logits, a key frame inside a clip == the same frame first, zero flow + identical frames => warped feature ==
    key feature, finite non-constant outputs."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W = 1024, 2048
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("101", H, W, demo_cfg)
    for k in list(arg):                       # FlowNet predictors -> exactly zero flow
        if k.startswith("Convolution") and arg[k].shape[0] == 2:
            arg[k] = np.zeros_like(arg[k])
    frames = synth.make_clip(H, W, 2)
    frames = [frames[0], frames[0].copy()]
    try:
        data = demo.build_batches(frames, demo_cfg)
        r = demo.ClipRunner("101", demo_cfg, arg, aux, (H, W))
        lg_k, lab_k = r.step(0, data[0], 2)
        a_logits, a_labels = lg_k.asnumpy().copy(), lab_k.asnumpy().copy()
        feat_key = r.feat.asnumpy().copy()
        lg_c, lab_c = r.step(1, data[1], 2)
        c_logits, c_labels = lg_c.asnumpy().copy(), lab_c.asnumpy().copy()
        np.testing.assert_allclose(r.feat.asnumpy(), feat_key, rtol=0, atol=1e-4 * float(np.abs(feat_key).max()))
        np.testing.assert_array_equal(c_labels[0], np.argmax(c_logits[0], axis=0))
        np.testing.assert_array_equal(a_labels[0], np.argmax(a_logits[0], axis=0))
        lg_k2, _ = r.step(0, data[0], 2)
        np.testing.assert_array_equal(lg_k2.asnumpy(), a_logits)
        lg_c2, _ = r.step(1, data[1], 2)
        np.testing.assert_array_equal(lg_c2.asnumpy(), c_logits)
        lg_k3, _ = r.step(2, data[1], 1)
        np.testing.assert_array_equal(lg_k3.asnumpy(), a_logits)
        # (seeded random weights give Accel-101 one dominant class at this size, so the label map may be constant:
        # the non-triviality checks are on the logits)
        assert np.isfinite(c_logits).all() and float(c_logits.std(axis=(2, 3)).min()) > 0
        # Accel-101's non-key frame is a different function of the frame than the key graph (feature fusion
        # 4096 -> 2048 on top of the warped feature): the two logit maps must not be trivially equal
        assert float(np.abs(c_logits - a_logits).max()) > 1e-3
    finally:
        tester.release_models()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("transport", ["cabi", "cabi-sendrecv", pytest.param("torch", marks=pytest.mark.gpu_extra)])
def test_config4_rccl_gather_on_one_gpu(demo_cfg, transport, monkeypatch):
    """FrameGather over RCCL with a world of one rank -- through accel_gather_logits of the C ABI ("cabi": libaccel_hip
    + librccl, the default) and through torch.distributed ("torch", the fallback): the gathered tensor of every frame
    must be byte-identical to the model's logits / labels buffer, across more frames than staging slots.
    "cabi-sendrecv": ACCEL_GATHER_SELF_SENDRECV=1 routes the root's own block through ncclGroupStart / ncclSend-to-self /
    ncclRecv-from-self / ncclGroupEnd, so the dlsym'd RCCL entry points, the dtype enum and the group semantics of the
    N > 1 path execute on hardware even though only one GPU is here."""
    import torch
    if transport == "cabi-sendrecv":
        monkeypatch.setenv("ACCEL_GATHER_SELF_SENDRECV", "1")
        transport = "cabi"
    import torch.distributed as dist
    from accel_amd import demo, dist as adist
    from accel_amd.core import tester
    H, W, interval = 128, 256, 3
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 4)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        data = demo.build_batches(frames, demo_cfg)
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        m = r.key_predictor._model
        g_logits = adist.FrameGather(m, m.ctx, "logits", (1, 19, H, W), "f4", 0, transport=transport)
        g_labels = adist.FrameGather(m, m.ctx, "labels", (1, H, W), "u1", 0, transport=transport)
        assert g_logits.transport == transport
        for t in range(4):
            lg, lab = r.step(t, data[t], interval)
            s0, s1 = g_logits.submit(), g_labels.submit()
            g_logits.drain()
            g_labels.drain()
            np.testing.assert_array_equal(g_logits.last(s0)[0].cpu().numpy(), lg.asnumpy())
            np.testing.assert_array_equal(g_labels.last(s1)[0].cpu().numpy(), np.uint8(lab.asnumpy()))
        g_logits.close()
        g_labels.close()
    finally:
        tester.release_models()
        dist.destroy_process_group()


@pytest.mark.parametrize("sendrecv", [False, True])
def test_config4_score_gather_expands_to_the_peers_own_logits(demo_cfg, sendrecv, monkeypatch):
    """The default payload of the multi-GPU gather (accel_gather_scores): the fused score maps of a frame travel, the root expands them.
    On one GPU (world of one; with ACCEL_GATHER_SELF_SENDRECV the block goes through ncclSend / ncclRecv to itself): over a key and
    three non-key frames at 1024x2048 the EXPANDED logits and labels must be bit-identical to the logits / labels the model itself
    left in its buffers, also in the image slots of an emulated second and third peer (accel_expand_scores on the communication stream)."""
    import torch
    import torch.distributed as dist
    from accel_amd import demo, dist as adist
    from accel_amd.core import tester
    if sendrecv:
        monkeypatch.setenv("ACCEL_GATHER_SELF_SENDRECV", "1")
    H, W, interval = 1024, 2048, 4
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 4)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        data = demo.build_batches(frames, demo_cfg)
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        m = r.key_predictor._model
        r.step(0, data[0], interval)                     # binds the key plan: the `scores` buffer exists from here on
        assert adist.ScoreGather.available(m)
        g = adist.ScoreGather(m, m.ctx, 1, H, W, 0, emulate_peers=2)
        assert g.transport == "cabi" and g.recv[0].shape == (1, 1, H // 16, W // 16, 20)
        for t in range(4):
            lg, lab = r.step(t, data[t], interval)
            g.submit((r.key_predictor if t % interval == 0 else r.cur_predictor).plan_for(H, W, 1)[0])
            g.drain()
            want, wlab = lg.asnumpy(), np.uint8(lab.asnumpy())
            got, glab = g.logits.cpu().numpy(), g.labels.cpu().numpy()
            for k in range(3):                           # the root's own slot and the two emulated peers'
                np.testing.assert_array_equal(got[k:k + 1], want, err_msg="frame %d, image slot %d" % (t, k))
                np.testing.assert_array_equal(glab[k:k + 1], wlab.reshape(1, H, W))
        g.close()
    finally:
        tester.release_models()
        dist.destroy_process_group()


def test_cabi_gather_without_torch_distributed(ctx):
    """accel_comm_* / accel_gather_logits stand-alone (a C host would do exactly this): one rank, the root's receive
    buffer must hold the send buffer's bytes; the send buffer may be overwritten right after the call returns."""
    import torch
    from accel_amd import runtime
    comm = runtime.Comm(ctx, 0, 1, runtime.Comm.unique_id())
    try:
        st = torch.cuda.ExternalStream(ctx.stream, device="cuda:0")
        with torch.cuda.stream(st):
            send = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
            recv = [torch.zeros_like(send) for _ in range(2)]
            for it in range(5):
                send.add_(1.0)                                   # compute-stream work producing the frame's output
                comm.gather(send.data_ptr(), recv[it & 1].data_ptr(), send.numel() * 4, 0)
            send.zero_()                                         # overwritten at once: staging holds what is in flight
        comm.sync()
        ctx.sync()
        base = torch.arange(1 << 20, dtype=torch.float32)
        assert torch.equal(recv[0].cpu(), base + 5.0) and torch.equal(recv[1].cpu(), base + 4.0)
    finally:
        comm.close()


def test_gather_beside_the_next_frames_compute_does_not_disturb_it(demo_cfg, monkeypatch):
    """The multi-GPU loop runs the RCCL transfer of frame t on the communication stream WHILE frame t+1 computes -- two
    hardware queues busy at once, the situation in which two-stream plans were seen to go wrong.  On the
    one GPU of a test box the root's own block is sent through ncclSend / ncclRecv to itself (ACCEL_GATHER_SELF_SENDRECV),
    frames are submitted back to back without a host wait in between, and every gathered frame must be bit-identical to
    the frame a run WITHOUT any gather produced: 1024x2048, 12 frames (key + 4 non-key, twice and a bit), logits."""
    import torch
    import torch.distributed as dist
    from accel_amd import demo, dist as adist
    from accel_amd.core import tester
    monkeypatch.setenv("ACCEL_GATHER_SELF_SENDRECV", "1")
    H, W, interval, n = 1024, 2048, 5, 12
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    frames = synth.make_clip(H, W, n)
    data = demo.build_batches(frames, demo_cfg)
    try:
        r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
        quiet = []
        for t in range(n):
            lg, _ = r.step(t, data[t], interval)
            quiet.append(lg.asnumpy()[0, :, ::3, ::3].copy())
        r.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
            m = r.key_predictor._model
            comm = adist.make_comm(m.ctx)
            src, nbytes = m.buffer("logits")[0], 19 * H * W * 4
            recv = [torch.empty((1, 19, H, W), dtype=torch.float32, device="cuda:0") for _ in range(n)]      # one buffer per frame: no host wait in the loop
            for t in range(n):
                r.step(t, data[t], interval)
                comm.gather(src, recv[t].data_ptr(), nbytes, 0)      # staged in compute-stream order, sent while frame t+1 computes
            comm.sync()
            got = [x.cpu().numpy()[0, :, ::3, ::3].copy() for x in recv]
            comm.close()
            r.close()
        finally:
            dist.destroy_process_group()
        for t in range(n):
            assert np.array_equal(got[t], quiet[t]), "frame %d gathered beside compute differs from the quiet run by %g" % (
                t, float(np.abs(got[t] - quiet[t]).max()))
    finally:
        tester.release_models()


def test_config5_accel50_f16_2048x4096(demo_cfg, monkeypatch):
    """Accel-50, fp16-MFMA convolutions with half activation storage, 2048x4096 (config 5's frame size), the first two frames of a
    kf=10 group (key, non-key: the chain through warp + the Accel-50 correction branch, accel_50.py:156-228): finite logits and
    non-degenerate label maps, and BOTH frames against the mode's own specification (below)."""
    from accel_amd import demo
    from accel_amd.core import tester
    H, W, interval = 2048, 4096, 10
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("50", H, W, demo_cfg)
    frames = synth.make_clip(H, W, 2)
    outs = {}
    monkeypatch.setenv("ACCEL_CONV_DTYPE", "f16")
    try:
        res = demo.run_clip("50", demo_cfg, arg, aux, frames, interval)
        outs["f16"] = [(lg[0][:, ::4, ::4].copy(), lab.copy()) for lg, lab in res]    # keep 1/16 of the logits
        del res
    finally:
        tester.release_models()
    for t, (b, lb) in enumerate(outs["f16"]):
        assert np.isfinite(b).all(), "frame %d: non-finite fp16-mode logits" % t
        assert len(np.unique(lb)) > 1
    # (rounds 3-4 also compared with the fp32 run of the same clip on the same path: 50 s of the suite for a weaker statement than the
    # one below; the fp32 path itself is checked against the oracle at this model in test_accel34_accel50_1024x2048_vs_oracle)
    # ... and against the mode's own SPECIFICATION at config 5's frame size: the oracle on half-rounded operands with the stored tensors
    # rounded once more (oracle.graphs ROUND_F16 + STORE_F16 = the layers the lowering stores as half), key frame + first non-key frame.
    # Two evaluations of ~100 discontinuous roundings decorrelate down to the half-precision noise (tests/test_f16_storage_gpu.py), so the
    # bar is statistical: typical pixel within 3e-4 of the logit range, worst pixel within 10 %, fewer than 0.5 % of the labels differ.
    from test_f16_storage_gpu import half_layers
    P = dict(arg)
    P.update(aux)
    G.ROUND_F16, G.STORE_F16 = True, half_layers("50", H, W, demo_cfg)
    try:
        # (a minute and a half of oracle time on 32 host threads: the key frame is the ResNet-101 graph, the non-key frame the branch
        # config 5 is named after; round 5 checked the key frame only)
        ref = G.run_clip(P, "50", _oracle_frames(frames[:2], demo_cfg), interval)
    finally:
        G.ROUND_F16, G.STORE_F16 = False, None
    for t, ((b, lb), (rlg, rlab)) in enumerate(zip(outs["f16"][:2], ref)):
        r = rlg[0][:, ::4, ::4]
        scale = max(1.0, float(np.abs(r).max()))
        d = np.abs(b - r).ravel() / scale
        mism = float((lb != rlab[0]).mean())
        print("config 5 (2048x4096, f16 + half storage) frame %d vs its specification: |error| / logit range median %.2e, 99.9 %% %.2e, max %.2e; "
              "labels differing %.4f %%" % (t, float(np.median(d)), float(np.quantile(d, 0.999)), float(d.max()), 100 * mism))
        assert float(np.median(d)) <= 3e-4 and float(d.max()) <= 0.1 and mism < 5e-3, "frame %d" % t


def test_headline_launch_geometries_are_replayed_not_timed(demo_cfg):
    """Reproducibility of the summation order: binding the headline workload (Accel-18, 1024x2048, one clip per call)
    takes every launch geometry from the shipped table -- no decision by timing -- and two independently bound
    models give bit-identical logits."""
    from accel_amd import demo, runtime
    from accel_amd.core import tester
    H, W = 1024, 2048
    demo_cfg.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, demo_cfg)
    data = demo.build_batches(synth.make_clip(H, W, 2), demo_cfg)
    outs, geo = [], []
    try:
        _, timed0, shipped = runtime.tune_stats()
        assert shipped > 0, "accel_amd/tune/gfx950.tune not found beside the library"
        for _ in range(2):
            r = demo.ClipRunner("18", demo_cfg, arg, aux, (H, W))
            outs.append([r.step(i, data[i], 2)[0].asnumpy().copy() for i in range(2)])
            geo.append([(o["name"], o["tile"], o["ksplit"]) for pred in (r.key_predictor, r.cur_predictor)
                        for o in pred.plan_for(H, W, 1)[0].ops() if o["kind"] == "conv"])
            r.close()
        replayed, timed1, _ = runtime.tune_stats()
        assert timed1 == timed0 and replayed > 0, "%d launch geometries were decided by timing" % (timed1 - timed0)
        assert geo[0] == geo[1], [(a, b) for a, b in zip(*geo) if a != b][:6]
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b)
    finally:
        tester.release_models()


def test_torch_initialises_after_the_library():
    """One process, libaccel_hip first, torch.cuda afterwards (the order of a script that only later sets up
    torch.distributed for the gather): both must work -- runtime.lib() makes the two share ONE HIP runtime."""
    import subprocess
    import sys
    code = ("import numpy as np\n"
            "from accel_amd import runtime\n"
            "ctx = runtime.Context(0)\n"
            "y = ctx.conv2d(np.ones((1, 8, 8, 8), np.float32), np.ones((8, 8, 3, 3), np.float32), None, 1, 1, 1)\n"
            "assert float(y[0, 0, 4, 4]) == 72.0\n"
            "import torch\n"
            "torch.cuda.set_device(0)\n"
            "t = torch.ones(4, device='cuda') * 2\n"
            "assert float(t.sum()) == 8.0\n"
            "y2 = ctx.conv2d(np.ones((1, 8, 8, 8), np.float32), np.ones((8, 8, 3, 3), np.float32), None, 1, 1, 1)\n"
            "assert float(y2[0, 0, 0, 0]) == 32.0\n"
            "print('both fine')\n")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "both fine" in out.stdout, out.stderr[-2000:]
