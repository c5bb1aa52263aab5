#!/usr/bin/env python
"""Headline benchmark: frames/s of the Accel-18 inference path on 1024x2048
clips at key-frame interval 5 (BASELINE.json metric, configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one clip per GPU = 1 key frame + 4 non-key frames through the HIP
path (key: ResNet-101-DCN + head; non-key: FlowNet-S, flow warp, ResNet-18-DCN
branch, two heads, fused upsample+correction+argmax).  Frames are synthetic,
weights seeded random (no checkpoints/data offline); the clip is resident in
HBM before the timed region; outputs (fp32 logits + uint8 labels) stay in HBM;
with N > 1 every frame's logits are gathered to rank 0 over RCCL, overlapped.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this driver

MFMA_F32_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
K80_ACCEL18_FPS = 1.0 / 0.44    # BASELINE.md section 1 (reference README.md:65): 0.44 s/frame on one Tesla K80


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--version", default="18")
    ap.add_argument("--size", default="1024x2048")
    ap.add_argument("--interval", type=int, default=5)
    ap.add_argument("--gather", default="logits", choices=["logits", "labels", "none"])
    ap.add_argument("--batch", type=int, default=int(os.environ.get("ACCEL_BENCH_BATCH", "8")),
                    help="clips processed together per GPU: every call runs one frame of each of B independent clips, the "
                         "convolutions see M = B*Ho*Wo (BASELINE config 4 shards 8 clips per GPU); 1 = the reference's batch")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("ACCEL_BENCH_LANES", "1")),
                    help="independent clip pipelines per GPU (own model, buffers and streams each)")
    ap.add_argument("--dtype", default=os.environ.get("ACCEL_CONV_DTYPE", "f32"), choices=["f32", "f16"],
                    help="f32 (default, the reference's precision: the headline) or f16 = fp16-MFMA convolutions with fp32 "
                         "storage/accumulate (BASELINE config 5; NOT the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(version, interval):
    """The CPU oracle (a port: the reference's MXNet cannot run here)
    timed on this box's host cores on a bounded sample."""
    from accel_amd.config.config import config
    from accel_amd.utils import image, synth
    from oracle import graphs as G
    h, w = 512, 1024
    arg, aux = synth.model_params(version, h, w, config)
    P = dict(arg)
    P.update(aux)
    fr = [image.transform(f, config.network.PIXEL_MEANS).astype(np.float32) for f in synth.make_clip(h, w, 2)]
    t0 = time.time()
    k = G.key_forward(P, fr[0])
    t1 = time.time()
    G.cur_forward(P, version, fr[1], fr[0], k["res5c_relu_output"])
    t2 = time.time()
    scale = (1024 * 2048) / float(h * w)
    per_frame = ((t1 - t0) + (interval - 1) * (t2 - t1)) / interval * scale
    return {"value": 1.0 / per_frame, "unit": "frames/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count())),
            "kind": "port",
            "sample": "oracle/ (C restatement, OpenMP) on 1 key + 1 non-key frame of Accel-%s at %dx%d "
                      "(1/%d of the pixels): key %.2f s, non-key %.2f s; kf=%d mean scaled x%d to 1024x2048"
                      % (version, h, w, int(scale), t1 - t0, t2 - t1, interval, int(scale))}


class _StdoutToStderr(object):
    """RCCL prints a version banner on the C stdout when a communicator comes up; the driver wants
    ONE JSON line on stdout, so fd 1 points at stderr while the process group / gathers run."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    a = parse()
    with _StdoutToStderr():
        out, finish = _run(a)
    if out is not None:
        print(json.dumps(out), flush=True)
    finish()


def _run(a):
    os.environ["ACCEL_CONV_DTYPE"] = a.dtype
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, a.gpus))
    H, W = [int(v) for v in a.size.split("x")]
    B = max(1, a.batch)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("ACCEL_BENCH_FORCE_DIST") == "1"   # exercise the RCCL gather path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from accel_amd import demo, dist as adist, runtime
    from accel_amd.config.config import config, update_config
    from accel_amd.core import tester
    from accel_amd.utils import image, synth
    update_config(os.path.join(HERE, "tests", "golden", "dff_deeplab_vid_demo.yaml"))
    config.SCALES[0] = (H, W)

    arg, aux = synth.model_params(a.version, H, W, config)
    # `lanes` independent clip pipelines per GPU: own model (weights, arena, buffers) and own HIP streams.
    # Lane l runs its clips rotated by l * interval / lanes frames, so a key frame of one lane overlaps the
    # non-key frames of the other (clips are independent: SURVEY.md 8e).
    lanes = []
    for l in range(max(1, a.lanes)):
        model = runtime.Model(runtime.Context(local_rank))
        runner = demo.ClipRunner(a.version, config, arg, aux, (H, W), context=[demo.mx.gpu(local_rank)], model=model, batch=B)
        key_plan, _ = runner.key_predictor.plan_for(H, W, B)
        cur_plan, _ = runner.cur_predictor.plan_for(H, W, B)
        lanes.append({"model": model, "key": key_plan, "cur": cur_plan, "gather": None, "rot": (l * a.interval) // max(1, a.lanes)})
    del arg, aux
    model, key_plan, cur_plan = lanes[0]["model"], lanes[0]["key"], lanes[0]["cur"]

    # B clips per step and lane, distinct per rank and clip, resident in HBM: dev_frames[t] = frame t of every clip
    clips = [synth.make_clip(H, W, a.interval, seed=20260929 + rank * 64 + b) for b in range(B)]
    dev_frames = [torch.from_numpy(np.concatenate([image.transform(c[t], config.network.PIXEL_MEANS).astype(np.float32) for c in clips], axis=0)).cuda()
                  for t in range(a.interval)]
    nbytes = B * 3 * H * W * 4

    gather_note = "none (single GPU)"
    if (world > 1 or force_dist) and a.gather != "none":
        try:
            for ln in lanes:
                if a.gather == "logits":
                    ln["gather"] = adist.FrameGather(ln["model"], ln["model"].ctx, "logits", (B, 19, H, W), "f4", local_rank)
                else:
                    ln["gather"] = adist.FrameGather(ln["model"], ln["model"].ctx, "labels", (B, H, W), "u1", local_rank)
            gather_note = "RCCL gather of per-frame %s to rank 0, async, double-buffered" % a.gather
        except Exception as e:   # keep the bench alive; the JSON says what happened
            for ln in lanes:
                ln["gather"] = None
            gather_note = "disabled: %r" % (e,)

    gather_failures = []

    def step():
        for i in range(a.interval):
            for ln in lanes:
                t = (i + ln["rot"]) % a.interval
                m = ln["model"]
                m.write_device("data", dev_frames[t].data_ptr(), nbytes)
                if t == 0:
                    ln["key"].run()
                else:
                    m.write_device("data_key", dev_frames[t - 1].data_ptr(), nbytes)
                    ln["cur"].run()
                if ln["gather"] is not None:
                    try:
                        ln["gather"].submit()
                    except Exception as e:      # a failing collective must not cost the whole measurement
                        ln["gather"] = None
                        gather_failures.append(repr(e))

    def sync():
        for ln in lanes:
            if ln["gather"] is not None:
                ln["gather"].drain()
            ln["model"].ctx.sync()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        frames_total = world * a.steps * a.interval * len(lanes) * B
        value = frames_total / elapsed
        out = {"metric": "frames/sec 1024x2048 Accel-%s kf=%d" % (a.version, a.interval), "value": round(value, 3),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": round(value / K80_ACCEL18_FPS, 2) if (a.version == "18" and (H, W) == (1024, 2048) and a.interval == 5 and a.dtype == "f32") else None,
               "baseline_note": "BASELINE.md 1: reference README 0.44 s/frame Accel-18 on 1x Tesla K80 (includes H2D + label D2H)",
               "dtype": "f32" if a.dtype == "f32" else "f16 operands on the matrix cores, f32 storage + accumulate (reduced precision: not the headline)",
               "data": "synthetic",
               "config": {"workload": "Accel-%s (R101-DCN key branch + FlowNet-S warp + R%s correction branch + fused score tail), "
                                      "%dx%d clips, key-frame interval %d, %d clip(s) (1 key + %d non-key frames each) per GPU per step, "
                                      "processed %d clips at a time (batched frames of independent clips)"
                                      % (a.version, a.version, H, W, a.interval, len(lanes) * B, a.interval - 1, B),
                          "frames_per_step_per_gpu": a.interval * len(lanes) * B, "clips_per_call": B, "clip_pipelines_per_gpu": len(lanes), "parallelism": "clip-sharded x%d (weights replicated)" % world,
                          "gather": gather_note + ("; DISABLED after failure: " + gather_failures[0] if gather_failures else ""), "weights": "seeded random",
                          "lowering": ("exact linear folds on (DESIGN.md 4): feat_upsampling*fc6 composed into one deconvolution; non-key L-head fc6 "
                                       "taken from the warped W_fc6*feat image of the key frame" if os.environ.get("ACCEL_FOLD_LINEAR", "1") != "0"
                                       else "reference layer list one to one (ACCEL_FOLD_LINEAR=0)"), "outputs": "fp32 logits 19xHxW + uint8 labels, left in HBM"}}
    if rank == 0 and not a.no_roofline:
        # dominant kernel = conv_igemm_f32 (implicit-GEMM conv on the fp32 matrix cores): HIP-event pair around
        # every launch on the compute stream, all conv launches of one clip (1 key + 4 non-key plans)
        kms, cms = key_plan.profile(2), cur_plan.profile(2)
        fl = ms = n = by = 0.0
        for plan, t, wgt in ((key_plan, kms, 1), (cur_plan, cms, a.interval - 1)):
            for op, d in zip(plan.ops(), t):
                if op["kind"] == "conv":
                    fl += wgt * op["flops"]
                    by += wgt * op["bytes"]
                    ms += wgt * float(d)
                    n += wgt
        clip_ms = float(kms.sum()) + (a.interval - 1) * float(cms.sum())
        ach = fl / (ms * 1e-3) / 1e12
        peak = MFMA_F32_PEAK_TFLOPS if a.dtype == "f32" else 2500.0     # dense fp16 MFMA peak, MI355X_MICROARCH.md
        # HBM bytes per launch come from the committed PMC passes of this same command (counters cannot be read
        # from inside the process): only reported for the workload they were collected on
        traffic, traffic_note = None, None
        tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tj) and a.version == "18" and (H, W) == (1024, 2048) and a.interval == 5 and a.dtype == "f32" and len(lanes) == 1:
            with open(tj) as f:
                tr = json.load(f)
        else:
            tr = None
        if tr is not None and int(tr.get("batch", 1)) == B:
            traffic = round(tr["read_bytes_per_launch"] + tr["write_bytes_per_launch"])
            traffic_note = "bytes per launch, profiles/r01_pmc_traffic.json: " + tr["method"]
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                           "algorithmic_bytes_per_launch": round(by / n),
                           "kernel": "conv_igemm_f32_kernel (all tile variants)", "launches_per_step": int(n),
                           "avg_launch_us": round(1e3 * ms / n, 2), "gflop_per_launch": round(fl / n / 1e9, 3),
                           "conv_ms_per_step": round(ms, 3), "all_kernels_ms_per_step": round(clip_ms, 3)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.version, a.interval)
    def finish():
        if dist is not None:
            with _StdoutToStderr():
                dist.barrier()
                dist.destroy_process_group()
    return (out if rank == 0 else None), finish


if __name__ == "__main__":
    main()
