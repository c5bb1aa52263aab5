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
Prints ONE JSON line on rank 0.  At N = 1 the same line carries `secondary` measurements taken with the same harness
(--secondary none skips them): Accel-101 (the other model BASELINE.json's metric names), Accel-18 at one clip per call
(the reference's TEST.BATCH_IMAGES: 1), and the PCIe-inclusive rate of the reference's own timing definition
(demo.py:234-250: host frame in, forward, label map back on the host) through page-locked, double-buffered transfers.
`vs_baseline` compares like with like: the PCIe-inclusive batch-1 rate against the reference's published K80 number.
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
    ap.add_argument("--gather", default="scores", choices=["scores", "logits", "labels", "none"],
                    help="N > 1: what every rank sends to rank 0 per frame.  scores (default): the fused score maps (0.66 MB per 1024x2048 frame), "
                         "expanded on the root into the same fp32 logits + labels bit for bit (accel_gather_scores; falls back to logits when the "
                         "model's upsampling filters are not uniform); logits: the 159 MB fp32 logits themselves (link-bound beyond ~480 frames/s "
                         "per GPU, DESIGN.md 6); labels: the uint8 label maps")
    ap.add_argument("--batch", type=int, default=8,
                    help="clips processed together per GPU: every call runs one frame of each of B independent clips, the "
                         "convolutions see M = B*Ho*Wo (BASELINE config 4 shards 8 clips per GPU); 1 = the reference's batch")
    ap.add_argument("--dtype", default=os.environ.get("ACCEL_CONV_DTYPE", "f32"), choices=["f32", "f16", "bf16x3"],
                    help="f32 (default, the reference's precision on the fp32 MFMA: the headline); bf16x3 = the same fp32 values with every "
                         "operand split exactly into three bf16 terms, six products on the bf16 matrix cores, fp32 accumulate "
                         "(fp32-equivalent results, tests/test_bf16x3_gpu.py); f16 = fp16-MFMA convolutions with fp32 storage/"
                         "accumulate (BASELINE config 5: reduced precision, never the headline)")
    ap.add_argument("--secondary", default="auto", choices=["auto", "none"],
                    help="auto: on a single GPU with the headline configuration also measure Accel-101, batch 1 and the "
                         "PCIe-inclusive loop (reported under `secondary`); none: headline only")
    ap.add_argument("--root-relief", type=int, default=-1,
                    help="N > 1 with --gather scores | logits: rank 0 (which also receives -- and with scores expands -- every other rank's frames) "
                         "processes this many clips fewer per call than the other ranks (dist.shard_clips(..., root_relief)); 0 = even shares; "
                         "default (-1): 1")
    ap.add_argument("--bind-inputs", action="store_true",
                    help="the plans read the resident frames where they lie (accel_model_bind_device) instead of copying them into the model's "
                         "input buffers: an extension, not the reference executor's semantics; reported as secondary.*_zero_copy_inputs by default")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: initialise torch.distributed anyway (exercises the RCCL gather path on one GPU)")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, have each print its rank / world size, exit (no GPU work)")
    ap.add_argument("--preflight", action="store_true",
                    help="N > 1 (always on there) or --force-dist: before any timed work every rank creates its communicator and sends one score map "
                         "(or frame) through the gather, under a watchdog that ends the job after --watchdog seconds and prints which rank did "
                         "not arrive; with this flag alone the job stops after the preflight and rank 0 prints its record")
    ap.add_argument("--watchdog", type=float, default=60.0, help="seconds a phase of a multi-rank job (communicator, preflight, one step) may take")
    ap.add_argument("--rank-logs", default=None, help="directory for the stdout of ranks > 0 (default: gpurun_out/ if it exists, else the temp directory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(version, interval, H=1024, W=2048):
    """The CPU oracle (a port: the reference's MXNet cannot run here) timed on this box's host cores on a
    bounded sample of the SAME workload: one key and one non-key frame at the full frame size (about 30 s)."""
    from accel_amd.config.config import config
    from accel_amd.utils import image, synth
    from oracle import graphs as G
    arg, aux = synth.model_params(version, H, W, config)
    P = dict(arg)
    P.update(aux)
    fr = [image.transform(f, config.network.PIXEL_MEANS).astype(np.float32) for f in synth.make_clip(H, W, 2)]
    t0 = time.time()
    k = G.key_forward(P, fr[0])
    t1 = time.time()
    G.cur_forward(P, version, fr[1], fr[0], k["res5c_relu_output"])
    t2 = time.time()
    per_frame = ((t1 - t0) + (interval - 1) * (t2 - t1)) / interval
    return {"value": round(1.0 / per_frame, 5), "unit": "frames/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count())),
            "kind": "port",
            "sample": "oracle/ (C restatement, OpenMP) on 1 key + 1 non-key frame of Accel-%s at %dx%d (the full frame size): "
                      "key %.2f s, non-key %.2f s; value = 1 / mean frame time of a kf=%d clip" % (version, H, W, t1 - t0, t2 - t1, interval)}


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


def _self_launch(a):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks ourselves (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) and hand their output through -- never a silent 1-GPU number."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, ACCEL_BENCH_SELF_LAUNCHED="1")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        sys.exit(_self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a number "
                         "for another job size" % (a.gpus, world))
    if a.launch_check:      # (tests: the launch path without a GPU) every rank reports itself and leaves
        _emit({"launch_check": True, "rank": int(os.environ.get("RANK", 0)), "world": world,
               "local_rank": int(os.environ.get("LOCAL_RANK", 0)), "self_launched": os.environ.get("ACCEL_BENCH_SELF_LAUNCHED") == "1"})
        return
    _rank_stdout_to_file(a, int(os.environ.get("RANK", 0)))
    with _StdoutToStderr():
        out, finish = _run(a)
    if out is not None:
        _emit(out)
    finish()


def _emit(record):
    """ONE write(2) call per record: the ranks of a job share the launcher's stdout pipe, and a line assembled from several writes
    (print: text, then the newline) can be cut in two by another rank's line (round-5 review: `JSONDecodeError: Extra data`)."""
    sys.stdout.flush()
    os.write(1, (json.dumps(record) + "\n").encode())


def _rank_stdout_to_file(a, rank):
    """Only rank 0 owns the job's stdout (the driver parses ONE JSON line from it): every other rank's fd 1 -- Python prints and
    the C libraries' alike -- goes to <--rank-logs>/bench_rank<r>.stdout for the whole run; stderr stays shared."""
    if rank == 0:
        return
    d = a.rank_logs or (os.path.join(HERE, "gpurun_out") if os.path.isdir(os.path.join(HERE, "gpurun_out")) else None)
    if d is None:
        import tempfile
        d = tempfile.gettempdir()
    try:
        os.makedirs(d, exist_ok=True)
        fd = os.open(os.path.join(d, "bench_rank%d.stdout" % rank), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    except OSError:
        fd = os.open(os.devnull, os.O_WRONLY)
    sys.stdout.flush()
    os.dup2(fd, 1)
    os.close(fd)


BF16_PEAK_TFLOPS = 2500.0       # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0          # HBM3E spec; HBM_ACHIEVABLE_GBPS is the guide's measured float4-copy rate
HBM_ACHIEVABLE_GBPS = 6300.0
HBM_REGIME_GBPS = 3000.0        # a conv launch that moves its ALGORITHMIC bytes at >= 3 TB/s is priced against HBM, not the matrix pipe

# executed matrix products per algorithmic multiply-add, dense peak of the pipe they run on, description
PIPES = {"conv_igemm_b3_kernel": (6.0, BF16_PEAK_TFLOPS, "bf16 MFMA, six products per fp32 multiply-add (three exact bf16 terms per operand): conv_igemm_b3 / conv_b3r / conv_b3d kernels"),
         "conv_wino_b3_kernel": (6.0 / 2.25, BF16_PEAK_TFLOPS, "bf16 MFMA, Winograd F(2x2,3x3) position GEMMs as six bf16 products each"),
         "conv_wino_f32_kernel": (1.0 / 2.25, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA, Winograd F(2x2,3x3): 16/36 of the direct multiply-adds"),
         "conv_igemm_f32_kernel": (1.0, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA"), "conv_stem_f32_kernel": (1.0, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA"),
         "conv1x1_ws_kernel": (1.0, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA (weight-stationary streaming 1x1)"),
         "conv_stem_b3_kernel": (6.0 * 176.0 / 147.0, BF16_PEAK_TFLOPS, "bf16 MFMA, the 7x7/2 stems as six bf16 products per multiply-add over K = 7 x 24 padded to 176 (147 algorithmic)"),
         "conv_stem_h2_kernel": (3.0 * 176.0 / 147.0, BF16_PEAK_TFLOPS, "fp16 MFMA, the 7x7/2 stems as three half products per multiply-add over K = 7 x 24 padded to 176 (147 algorithmic): conv_stem_b3<H2>"),
         "conv_halo_h2_kernel": (3.0, BF16_PEAK_TFLOPS, "fp16 MFMA, THREE products per fp32 multiply-add: conv_halo.hip (geometry 78: 3x3 layers with few output channels, the patch staged once with its halo)"),
         "conv_h2_kernel": (3.0, BF16_PEAK_TFLOPS, "fp16 MFMA (same dense peak as bf16), THREE products per fp32 multiply-add (two half terms per operand, hi*hi + hi*lo + lo*hi): conv_b3r<NPL=2>"),
         "conv_wino_h2_kernel": (3.0 / 2.25, BF16_PEAK_TFLOPS, "fp16 MFMA, Winograd F(2x2,3x3) position GEMMs as three half products each (conv_wino_b3 / conv_wino_b3s, H2)"),
         "conv_f16_kernel": (1.0, BF16_PEAK_TFLOPS, "fp16 MFMA, ONE half product per multiply-add (f16-mode layer on any geometry: conv_igemm_f16 / conv_b3r<NPL=1> / conv_b3d<NPL=1>)")}


def conv_family(op, dtype="f32"):
    """Kernel family of one convolution launch.  `mode` (accel_plan_op_mode) is the arithmetic of the LAYER: an f16-mode layer
    executes one half product per multiply-add on whichever geometry it was given (70-87 included), an fp32 layer on geometries
    70-87 executes the six products of the three-term bf16 split -- unless the layer is in its fp16x2 form (mode 3, the default), where
    geometries 76-81, 41-43 and the stem (51) execute THREE half products (70-75 and 82-87 keep the bf16x3 form)."""
    t, mode = op["tile"], op.get("mode", 1 if dtype == "f16" else 2 if dtype == "bf16x3" else 0)
    if op.get("narrow"):
        return "conv_narrow_kernel"
    if mode == 1:
        return "conv_f16_kernel"
    if t in (41, 42, 43):
        return "conv_wino_h2_kernel" if mode == 3 else "conv_wino_b3_kernel"
    if mode == 3 and t == 78:
        return "conv_halo_h2_kernel"
    if mode == 3 and 76 <= t <= 81:
        return "conv_h2_kernel"
    if t == 40:
        return "conv_wino_f32_kernel"
    if t == 50:
        return "conv_stem_f32_kernel"
    if t == 51:
        return "conv_stem_h2_kernel" if mode == 3 else "conv_stem_b3_kernel"
    if t == 60:
        return "conv1x1_ws_kernel"
    if mode == 2 or 70 <= t <= 87:
        return "conv_igemm_b3_kernel"
    return "conv_igemm_f32_kernel"


def roofline_from_launches(launches, dtype="f32"):
    """launches: [(op dict of Plan.ops(), HIP-event duration in ms, launches per step)].

    Every convolution launch is classed by REGIME first: one whose algorithmic bytes / duration reach HBM_REGIME_GBPS is bound by
    HBM (the short-K 1x1 layers of res2 / res3 and the 64-channel layers) and is priced as bytes/s against the HBM roof; the rest
    are priced against the matrix pipe their family EXECUTES on (PIPES: bf16x3 = 6 bf16 products per multiply-add at 2500 TFLOP/s,
    Winograd 16/36 of the multiply-adds, fp16 mode ONE product).  The headline object is the deep-K (matrix-regime) class of the
    dominant family; `hbm_class` is the same report for the HBM-regime launches; `all_conv.frac` = the share of the matrix-regime
    convolution time an ideal pipe would need."""
    fam, tot = {}, {"mfma": [0.0, 0.0, 0.0, 0.0, 0.0], "hbm": [0.0, 0.0, 0.0, 0.0, 0.0]}     # launches, ms, flops, bytes, ideal ms
    cached = [0.0, 0.0, 0.0, []]      # launches, ms, bytes, names: algorithmic bytes / duration ABOVE the HBM peak -- served by the caches, never priced as HBM
    fl = ms = n = by = clip_ms = 0.0
    for op, d, wgt in launches:
        clip_ms += wgt * d
        if op["kind"] != "conv" or d <= 0:
            continue
        fl += wgt * op["flops"]; by += wgt * op["bytes"]; ms += wgt * d; n += wgt
        name = conv_family(op, dtype)
        gbps = op["bytes"] / (d * 1e-3) / 1e9
        if gbps > HBM_PEAK_GBPS:
            # no launch may be priced above what the part has: its operands came out of the Infinity Cache / L2 (small maps, weights of the
            # previous launch), so it belongs to neither roof's class; reported separately, with names, so that a mis-counted `bytes`
            # (round 4: the stride-2 1x1 layers counted the three quarters of the input they never read) shows up here instead of
            # inflating hbm_class.achieved
            cached[0] += wgt; cached[1] += wgt * d; cached[2] += wgt * op["bytes"]
            cached[3].append("%s (%.0f GB/s)" % (op.get("name", "?"), gbps))
            continue
        regime = "hbm" if (gbps >= HBM_REGIME_GBPS or name not in PIPES) else "mfma"
        f = fam.setdefault(name, {"mfma": [0.0, 0.0, 0.0, 0.0], "hbm": [0.0, 0.0, 0.0, 0.0]})[regime]
        f[0] += wgt; f[1] += wgt * d; f[2] += wgt * op["flops"]; f[3] += wgt * op["bytes"]
        t = tot[regime]
        t[0] += wgt; t[1] += wgt * d; t[2] += wgt * op["flops"]; t[3] += wgt * op["bytes"]
        if name in PIPES:
            mul, pk, _ = PIPES[name]
            t[4] += mul * wgt * op["flops"] / (pk * 1e12) * 1e3

    def agg(v, name):
        out = {"launches_per_step": int(v[0]), "ms_per_step": round(v[1], 3)}
        if v[0]:
            alg = v[2] / (v[1] * 1e-3) / 1e12
            out.update({"avg_launch_us": round(1e3 * v[1] / v[0], 2), "algorithmic_tflops": round(alg, 2),
                        "hbm_gbps_algorithmic": round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                        "algorithmic_bytes_per_launch": round(v[3] / v[0]), "algorithmic_gflop_per_launch": round(v[2] / v[0] / 1e9, 3)})
            if name in PIPES:
                mul, pk, _ = PIPES[name]
                out.update({"executed_tflops": round(mul * alg, 1), "executed_peak_tflops": pk, "executed_frac": round(mul * alg / pk, 4)})
        return out
    families = {}
    for k, v in sorted(fam.items()):
        both = [v["mfma"][i] + v["hbm"][i] for i in range(4)]
        d = agg(both, k)
        d["pipe"] = PIPES[k][2] if k in PIPES else "no matrix instructions (strip kernel for the 2-channel flow predictors): bandwidth / latency bound"
        d["matrix_regime"] = agg(v["mfma"], k)
        d["hbm_regime"] = agg(v["hbm"], k)
        families[k] = d
    dom = max((k for k in families if k in PIPES), key=lambda k: families[k]["matrix_regime"]["ms_per_step"])
    D = families[dom]["matrix_regime"]
    hb = tot["hbm"]
    hbm_gbps = hb[3] / (hb[1] * 1e-3) / 1e9 if hb[1] else 0.0
    mf = tot["mfma"]
    return {"bound": "mfma", "kernel": dom, "achieved": D.get("executed_tflops"), "peak": D.get("executed_peak_tflops"), "unit": "TFLOP/s",
            "frac": D.get("executed_frac"), "traffic": None, "traffic_note": None,
            "achieved_note": "the DEEP-K (matrix-regime) launches of the dominant convolution family (%s: %.1f of %.1f ms of convolution time per step): "
                             "EXECUTED flops (algorithmic 2 x MAC x the products its geometry executes per multiply-add) / the sum of their HIP-event "
                             "durations, against the dense peak of the pipe it runs on (%s); launches that move their algorithmic bytes at >= %.0f GB/s "
                             "are reported under hbm_class instead" % (dom, D["ms_per_step"], ms, PIPES[dom][2], HBM_REGIME_GBPS),
            "avg_launch_us": D.get("avg_launch_us"), "launches_per_step": D["launches_per_step"],
            "algorithmic_tflops": D.get("algorithmic_tflops"), "algorithmic_bytes_per_launch": D.get("algorithmic_bytes_per_launch"),
            "algorithmic_gflop_per_launch": D.get("algorithmic_gflop_per_launch"),
            "hbm_class": {"bound": "hbm", "achieved": round(hbm_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_gbps / HBM_PEAK_GBPS, 4),
                          "frac_of_achievable": round(hbm_gbps / HBM_ACHIEVABLE_GBPS, 4), "launches_per_step": int(hb[0]), "ms_per_step": round(hb[1], 3),
                          "what": "convolution launches whose algorithmic bytes (input + weights + residual + output, once each) / duration reach %.0f GB/s: "
                                  "priced as bytes/s against HBM (8 TB/s spec, %.1f TB/s achievable per MI355X_MICROARCH.md)" % (HBM_REGIME_GBPS, HBM_ACHIEVABLE_GBPS / 1e3)},
            "cache_class": {"launches_per_step": int(cached[0]), "ms_per_step": round(cached[1], 3), "layers": cached[3][:12],
                            "what": "convolution launches whose algorithmic bytes / duration EXCEED the %.0f GB/s HBM peak: operands served by the caches; "
                                    "excluded from hbm_class and from the families' rates" % HBM_PEAK_GBPS},
            "all_conv": {"frac": round(mf[4] / mf[1], 4) if mf[1] else None,
                         "what": "time-weighted over every MATRIX-regime convolution launch of a step: sum(executed flop_i / peak_i) / sum(t_i)",
                         "matrix_regime_ms_per_step": round(mf[1], 3), "hbm_regime_ms_per_step": round(hb[1], 3),
                         "algorithmic_tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms else None, "launches_per_step": int(n), "avg_launch_us": round(1e3 * ms / n, 2) if n else None,
                         "gflop_per_launch": round(fl / n / 1e9, 3) if n else None, "algorithmic_bytes_per_launch": round(by / n) if n else None,
                         "conv_ms_per_step": round(ms, 3), "all_kernels_ms_per_step": round(clip_ms, 3)},
            "families": families}


class Workload(object):
    """Accel-<version> on this GPU, B clips per call, the clips (interval frames each) resident in HBM."""

    def __init__(self, version, B, H, W, interval, local_rank, rank, config):
        import torch
        from accel_amd import demo, runtime
        from accel_amd.utils import image, synth
        self.version, self.B, self.H, self.W, self.interval = str(version), B, H, W, interval
        self.config, self.local_rank, self.torch = config, local_rank, torch
        self.arg, self.aux = synth.model_params(version, H, W, config)
        self.model = runtime.Model(runtime.Context(local_rank))
        self.runner = demo.ClipRunner(version, config, self.arg, self.aux, (H, W), context=[demo.mx.gpu(local_rank)],
                                      model=self.model, batch=B)
        self.key, _ = self.runner.key_predictor.plan_for(H, W, B)
        self.cur, _ = self.runner.cur_predictor.plan_for(H, W, B)
        self.cur_b, _ = self.runner.cur_predictor.plan_for(H, W, B, slot=1)     # ping-pong partner (the same plan if unpaired)
        # B clips, distinct per rank and clip: frames[t] = frame t of every clip, (B, 3, H, W) fp32, mean-subtracted
        clips = [synth.make_clip(H, W, interval, seed=20260929 + rank * 64 + b) for b in range(B)]
        self.host_frames = [np.concatenate([image.transform(c[t], config.network.PIXEL_MEANS).astype(np.float32) for c in clips], axis=0)
                            for t in range(interval)]
        self.dev_frames = [torch.from_numpy(f).cuda() for f in self.host_frames]
        self.nbytes = B * 3 * H * W * 4
        self.gather = None
        self.bind_inputs = False      # --bind-inputs

    @property
    def scores_gather(self):
        return type(self.gather).__name__ == "ScoreGather"

    def step(self):
        """one clip per lane: key frame + (interval - 1) non-key frames, inputs and outputs in HBM"""
        m = self.model
        # the frames are resident in HBM and are COPIED into the model's input buffers every frame, as the reference's executor does
        # with a source array on the device (executor_group._load_general: d_src.copyto(d_targets), unconditionally); the headline
        # includes those copies.  --bind-inputs: the plans read the frames where they lie (accel_model_bind_device) --
        # reported as secondary.*_zero_copy_inputs, never as the headline
        put = m.bind_device if self.bind_inputs else m.write_device
        for t in range(self.interval):
            put("data", self.dev_frames[t].data_ptr(), self.nbytes)
            if t == 0:
                plan = self.key
            else:
                put("data_key", self.dev_frames[t - 1].data_ptr(), self.nbytes)
                plan = self.cur if t % 2 else self.cur_b      # frame 1 reads the key plan's `feat`, frame 2 `feat_b`, ...
            plan.run()
            if self.gather is not None:
                self.gather.submit(plan) if self.scores_gather else self.gather.submit()

    def sync(self):
        if self.gather is not None:
            self.gather.drain()
        self.model.ctx.sync()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup, dist=None):
        for _ in range(warmup):
            self.step()
        self.sync()
        if dist is not None:
            dist.barrier()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.sync()
        self.own_elapsed = time.perf_counter() - t0      # this rank's own work, before it waits for the others
        if dist is not None:
            dist.barrier()
        self.sync()
        return time.perf_counter() - t0

    def timed_pcie(self, steps, warmup):
        """The reference's timing definition (demo.py:234-250): per frame, the image comes from host memory, the
        forward runs, the label map arrives on the host -- through the Predictor / im_segment surface, with page-locked
        frames and the NEXT frame's upload started beside the running forward (Predictor.prefetch)."""
        from accel_amd import mx
        ctx = mx.cpu_pinned()
        arrs = [mx.nd.array(f, ctx=ctx) for f in self.host_frames]
        zero = mx.nd.array(np.zeros((self.B, 2048, 1, 1), np.float32))
        batches = [[arrs[t], arrs[t - 1] if t else arrs[0], zero] for t in range(self.interval)]
        n = self.interval

        def clip():
            for t in range(n):
                _, lab = self.runner.step(t, batches[t], n)
                self.runner.prefetch(batches[(t + 1) % n])
                host = lab.asnumpy()
            return host
        for _ in range(warmup):
            clip()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = clip()
        self.sync()
        el = time.perf_counter() - t0
        assert last.shape == (self.B, self.H, self.W)
        return el

    def conv_roofline(self, dtype):
        """Roofline of the convolution kernels: a HIP-event pair around every launch on the compute stream
        (accel_plan_profile), all conv launches of one step (1 key + interval-1 non-key plans); see roofline_from_launches."""
        kms, cms = self.key.profile(2), self.cur.profile(2)
        launches = []
        for plan, t, wgt in ((self.key, kms, 1), (self.cur, cms, self.interval - 1)):
            launches += [(op, float(d), wgt) for op, d in zip(plan.ops(), t)]
        return roofline_from_launches(launches, dtype)

    def close(self):
        if self.gather is not None:
            try:
                self.gather.close()
            except Exception:
                pass
        self.model.close()
        self.model.ctx.close()
        self.dev_frames = None


def _pmc_traffic(version, H, W, interval, dtype, B):
    """HBM bytes per launch of the DOMINANT convolution family (the bf16x3 implicit GEMM) from the committed PMC passes of
    this same command (counters cannot be read from inside the process): only reported for the workload they were collected on."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "r*_pmc_traffic.json")))
    if not files or not (version == "18" and (H, W) == (1024, 2048) and interval == 5 and dtype == "f32"):
        return None, None, []
    with open(files[-1]) as f:
        tr = json.load(f)
    if int(tr.get("batch", 1)) != B:
        return None, None, []
    d = tr.get("dominant", tr)
    return (round(d["read_bytes_per_launch"] + d["write_bytes_per_launch"]),
            "HBM read + write bytes per launch of %s from the committed PMC passes (%s): %s"
            % (" / ".join(d.get("kernels", ["the convolution kernels"])), os.path.relpath(files[-1], HERE), tr["method"]),
            d.get("kernels", []))


# rocprof kernel name -> the families of roofline_from_launches its launches fall into
PMC_KERNEL_FAMILIES = {"conv_b3r_kernel": ("conv_h2_kernel", "conv_igemm_b3_kernel", "conv_f16_kernel"), "conv_igemm_b3_kernel": ("conv_igemm_b3_kernel",),
                       "conv_wino_b3_kernel": ("conv_wino_h2_kernel", "conv_wino_b3_kernel"), "conv_wino_b3s_kernel": ("conv_wino_h2_kernel", "conv_wino_b3_kernel")}


def pair_traffic(roof, traffic, note, kernels):
    """roofline.traffic belongs to a SET of launches (every launch of the kernels the PMC passes sampled): it is paired with the
    algorithmic bytes of the same set -- all regimes of the families those kernels carry -- and the ratio is printed."""
    roof["traffic"], roof["traffic_note"] = traffic, note
    if traffic is None:
        return roof
    fams = sorted(set(f for k in kernels for f in PMC_KERNEL_FAMILIES.get(k, ()) if f in roof["families"]))
    n = sum(roof["families"][f]["launches_per_step"] for f in fams)
    by = sum(roof["families"][f].get("algorithmic_bytes_per_launch", 0) * roof["families"][f]["launches_per_step"] for f in fams)
    if n:
        roof["traffic_algorithmic_bytes_per_launch"] = round(by / n)
        roof["traffic_ratio"] = round(traffic / (by / n), 3)
        roof["traffic_set"] = "all %d launches per step of the families %s (matrix and HBM regime alike) -- NOT the deep-K class `achieved` is quoted on" % (n, ", ".join(fams))
    return roof


def _gather_self_secondary(wl, a, B, H, W, local_rank, steps, warm, rate):
    """What the collective costs the COMPUTE it runs beside, measured on the one GPU a bench box has: the headline loop with a
    world-of-one RCCL communicator whose root block travels through ncclSend / ncclRecv to itself (ACCEL_GATHER_SELF_SENDRECV=1),
    i.e. RCCL kernels moving the full per-call payload on the communication queue beside the next frame's kernels.  The link itself
    (one xGMI hop per peer) is not exercised: DESIGN.md 6 combines these rates with the per-link limit."""
    out = {}
    import torch
    import torch.distributed as dist
    from accel_amd import dist as adist
    made = False
    saved = os.environ.get("ACCEL_GATHER_SELF_SENDRECV")
    try:
        if not dist.is_initialized():
            import socket
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            made = True
        os.environ["ACCEL_GATHER_SELF_SENDRECV"] = "1"
        for what, shape, dt, per_frame in (("scores", None, None, (H // 16) * (W // 16) * 20 * 4), ("scores_root_of_8", None, None, (H // 16) * (W // 16) * 20 * 4),
                                           ("logits", (B, 19, H, W), "f4", 19 * H * W * 4), ("labels", (B, H, W), "u1", H * W)):
            name = "accel18_batch%d_gather_self_%s" % (B, what)
            try:
                if what.startswith("scores"):
                    if not adist.ScoreGather.available(wl.model):
                        out[name] = {"error": "the model has no `scores` buffer (upsampling filters not uniform)"}
                        continue
                    wl.gather = adist.ScoreGather(wl.model, wl.model.ctx, B, H, W, local_rank, emulate_peers=7 if what == "scores_root_of_8" else 0)
                else:
                    wl.gather = adist.FrameGather(wl.model, wl.model.ctx, what, shape, dt, local_rank, transport="cabi")
                el = wl.timed(steps, warm)
                v = rate(wl, el, steps)
                out[name] = {"value": v, "unit": "frames/s", "payload_bytes_per_call": per_frame * B,
                             "peer_link_gbps_at_this_rate": round(per_frame * v / 1e9, 2),
                             "what": ("headline loop + per-frame gather of the fused score maps through ncclSend / ncclRecv to self and their expansion into "
                                      "fp32 logits + labels on the communication stream: what a PEER pays" if what == "scores" else
                                      "the same with the root's work of an 8-GPU job: besides its own block the expansion of SEVEN more blocks per call "
                                      "(%.1f GB of logits written per call on the communication stream beside the next frame's kernels): what the ROOT pays"
                                      % (7 * B * 19 * H * W * 4 / 1e9) if what == "scores_root_of_8" else
                                      "headline loop + per-frame gather of %s through ncclSend / ncclRecv to self on the communication stream "
                                      "(compute-side cost of the collective)" % what) +
                                     "; peer_link_gbps_at_this_rate = what ONE peer's xGMI link to the root would have to carry at this frame rate"}
            except Exception as e:
                out[name] = {"error": repr(e)}
            finally:
                if wl.gather is not None:
                    try:
                        wl.sync()
                        wl.gather.close()
                    except Exception:
                        pass
                wl.gather = None
    except Exception as e:
        out["accel18_batch%d_gather_self_logits" % B] = {"error": repr(e)}
    finally:
        if saved is None:
            os.environ.pop("ACCEL_GATHER_SELF_SENDRECV", None)
        else:
            os.environ["ACCEL_GATHER_SELF_SENDRECV"] = saved
        if made:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    return out


def _run(a):
    os.environ["ACCEL_CONV_DTYPE"] = a.dtype
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, a.gpus))
    H, W = [int(v) for v in a.size.split("x")]
    B = max(1, a.batch)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = a.force_dist   # exercise the RCCL gather path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from accel_amd import dist as adist
    from accel_amd.config.config import config, update_config
    update_config(os.path.join(HERE, "tests", "golden", "dff_deeplab_vid_demo.yaml"))
    config.SCALES[0] = (H, W)

    # the root of the logits gather also receives every other rank's frames: --root-relief gives it that many clips fewer per call
    relief = min(a.root_relief if a.root_relief >= 0 else 1, B - 1) if (world > 1 and a.gather in ("logits", "scores")) else 0
    B_rank = B - relief if rank == 0 else B
    wl = Workload(a.version, B_rank, H, W, a.interval, local_rank, rank, config)
    wl.bind_inputs = a.bind_inputs

    gather_note = "none (single GPU)"
    payload = a.gather
    if (world > 1 or force_dist) and a.gather != "none":
        try:
            if payload == "scores":
                # every rank looks at its own model and the ranks agree (the models are replicas; a disagreement must not split the job)
                ok = torch.tensor([1 if adist.ScoreGather.available(wl.model) else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) != 1:
                    payload = "logits"
            if payload == "scores":
                wl.gather = adist.ScoreGather(wl.model, wl.model.ctx, B, H, W, local_rank, own_images=B_rank)
            elif payload == "logits":
                wl.gather = adist.FrameGather(wl.model, wl.model.ctx, "logits", (B, 19, H, W), "f4", local_rank,
                                              own_bytes=B_rank * 19 * H * W * 4 if B_rank != B else None)
            else:
                wl.gather = adist.FrameGather(wl.model, wl.model.ctx, "labels", (B, H, W), "u1", local_rank)
            note = getattr(wl.gather, "transport_note", "")
            gather_note = "RCCL gather of per-frame %s to rank 0, async, double-buffered, transport %s%s" % (
                payload + (" (fused score maps, expanded on the root into fp32 logits + labels, bit-identical)" if payload == "scores" else
                           " (asked for scores: the model has no `scores` buffer)" if a.gather == "scores" else ""),
                wl.gather.transport, ("; " + note) if note else "")
        except Exception as e:   # keep the bench alive; the JSON says what happened
            wl.gather = None
            gather_note = "disabled: %r" % (e,)
        # ... on EVERY rank or on none: a rank that dropped its gather alone would leave the others in ncclSend / ncclRecv groups it never
        # joins (round-5 review, weak 7a).  The constructors decide collectively already; this vote covers what remains (an allocation
        # that failed on one rank).
        if adist._vote_any(wl.gather is None, wl.model.ctx, None) and wl.gather is not None:
            wl.gather.close()
            wl.gather = None
            gather_note = "disabled: another rank could not set its gather up"

    # Multi-rank jobs fail FAST and the same way on every rank: each phase has a deadline (accel_amd.dist.Watchdog: the rank that gives
    # up prints which ranks never reached the phase), and a collective that raises ends the rank -- torch.distributed.run then ends
    # the job -- instead of being dropped by one rank while the others keep issuing it.
    wd = adist.Watchdog(rank, world, adist.default_store()) if dist is not None else None
    preflight = None
    if wd is not None and wl.gather is not None:
        t0 = time.perf_counter()
        with wd.phase("preflight gather", a.watchdog):
            if wl.scores_gather:
                wl.gather.submit(wl.key)      # the maps of a plan that has not run yet: whatever the `scores` buffer holds travels
            else:
                wl.gather.submit()
            wl.gather.drain()
            wl.sync()
        preflight = {"ok": True, "seconds": round(time.perf_counter() - t0, 3), "payload": payload, "transport": wl.gather.transport}
    if a.preflight:
        if wd is not None:
            wd.close()
        out = {"preflight": preflight or {"ok": wl.gather is not None, "note": gather_note}, "n_gpus": world, "gather": gather_note} if rank == 0 else None
        wl.close()

        def finish_pre():
            if dist is not None:
                with _StdoutToStderr():
                    dist.barrier()
                    dist.destroy_process_group()
        return out, finish_pre

    gather_failures = []
    _step = wl.step

    def guarded_step():
        if wd is not None:
            wd.beat()
        try:
            _step()
        except Exception as e:
            if wl.gather is None or world > 1:
                # N > 1: no rank may carry on without a collective the others still issue -- say what happened and end the job
                sys.stderr.write("bench.py rank %d of %d: step failed (%r); ending the job\n" % (rank, world, e))
                sys.stderr.flush()
                raise
            # one rank (--force-dist on a single GPU): the measurement goes on without the gather and the record says so
            wl.gather = None
            gather_failures.append(repr(e))
            _step()
    wl.step = guarded_step

    if wd is not None:
        with wd.phase("timed steps", max(a.watchdog, 60.0)):
            elapsed = wl.timed(a.steps, a.warmup, dist)
    else:
        elapsed = wl.timed(a.steps, a.warmup, dist)
    rank_ms = None
    if dist is not None:
        # every rank's own time for the K steps (between the two barriers a rank that finishes early waits in the second one, so
        # its own clock is stopped before that barrier): a straggling root shows here
        own = torch.tensor([wl.own_elapsed], dtype=torch.float64, device="cuda")
        alls = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(alls, own)
        rank_ms = [round(1e3 * float(x.item()) / a.steps, 3) for x in alls]
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: the same steps once more with the OTHER payload of the gather (uint8 label maps are 76x smaller than fp32
    # logits), so the line shows what the collective costs; and the rate each peer's xGMI link to the root carries
    gather_detail = None
    if world > 1 and a.gather in ("scores", "logits", "labels") and wl.gather is not None and not gather_failures:
        other = "logits" if payload != "logits" else "labels"
        try:
            wl.sync()
            wl.gather.close()
            wl.gather = (adist.FrameGather(wl.model, wl.model.ctx, "labels", (B, H, W), "u1", local_rank,
                                           own_bytes=B_rank * H * W if B_rank != B else None) if other == "labels"
                         else adist.FrameGather(wl.model, wl.model.ctx, "logits", (B, 19, H, W), "f4", local_rank,
                                                own_bytes=B_rank * 19 * H * W * 4 if B_rank != B else None))
            with wd.phase("timed steps, other payload", max(a.watchdog, 60.0)):
                el2 = wl.timed(a.steps, 1, dist)
            t2 = torch.tensor([el2], dtype=torch.float64, device="cuda")
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            el2 = float(t2.item())
            per_frame = {"logits": 19 * H * W * 4, "labels": H * W, "scores": (H // 16) * (W // 16) * 20 * 4}
            fps = lambda el: (world * B - relief) * a.steps * a.interval / el
            gather_detail = {
                "gather_" + payload: {"value": round(fps(elapsed), 2), "unit": "frames/s",
                                      "peer_link_gbps": round(per_frame[payload] * fps(elapsed) / world / 1e9, 2)},
                "gather_" + other: {"value": round(fps(el2), 2), "unit": "frames/s",
                                    "peer_link_gbps": round(per_frame[other] * fps(el2) / world / 1e9, 2)},
                "note": "every peer sends its frames over its own direct xGMI link to rank 0 (about 153 GB/s per link and direction); "
                        "peer_link_gbps = payload bytes per frame x frames/s of one rank; the root receives (N - 1) x that"}
        except Exception as e:
            gather_detail = {"error": repr(e)}

    headline_cfg = a.version == "18" and (H, W) == (1024, 2048) and a.interval == 5 and a.dtype == "f32"
    out = None
    if rank == 0:
        frames_total = (world * B - relief) * a.steps * a.interval
        value = frames_total / elapsed
        out = {"metric": "frames/sec 1024x2048 Accel-%s kf=%d" % (a.version, a.interval), "value": round(value, 3),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "baseline_note": "BASELINE.md 1: reference README 0.44 s/frame Accel-18 on 1x Tesla K80, batch 1, including H2D of the "
                                "frame and D2H of the label map; vs_baseline = secondary.accel18_batch1_pcie_inclusive / that number "
                                "(same timing definition, other hardware), null when that secondary was not measured",
               "dtype": (("f32 (storage, accumulation and results; the launch geometries the tuner picks include the fp32 MFMA, Winograd F(2x2,3x3) "
                          "and, for most layers, 'fp16x2': each fp32 operand as TWO half terms hi + lo (22-23 significant bits, operands "
                          "centred in the half range by exact powers of two: weights per channel, pixels by a scale each convolution derives from the largest "
                          "|value| of its input tensor IN THE SAME RUN -- no calibration, a run is a pure function of its inputs), "
                          "three fp16 MFMA products hi*hi + hi*lo + lo*hi accumulated in fp32 -- error against float64 equal to the "
                          "bf16x3 form's and below an fp32 accumulation's, tests/test_h2_gpu.py, tests/test_stateless_gpu.py; secondary.*_bf16x3_split = the "
                          "three-term bf16 form (six products), secondary.*_fp32_mfma_only = neither)"
                          if os.environ.get("ACCEL_SPLIT", "h2") == "h2" else
                          "f32 (storage, accumulation and results; the launch geometries the tuner picks include the fp32 MFMA, Winograd F(2x2,3x3) "
                          "and, for most implicit-GEMM layers, 'bf16x3': each fp32 operand split EXACTLY into three bf16 terms, six bf16 "
                          "MFMA products accumulated in fp32 -- error against float64 equal to the fp32-MFMA kernel's, "
                          "tests/test_bf16x3_gpu.py; secondary.*_fp32_mfma_only = without them)")
                         if "split" not in os.environ.get("ACCEL_WITHHOLD", "").split(",") else "f32 (fp32 MFMA only: ACCEL_WITHHOLD=split)") if a.dtype == "f32" else
                        "f32 as 3 x bf16 (each fp32 operand split exactly into three bf16 terms, six products per multiply-add on the bf16 "
                        "matrix cores, f32 storage + accumulate; error vs float64 equal to the fp32-MFMA kernel's, tests/test_bf16x3_gpu.py)"
                        if a.dtype == "bf16x3" else
                        "f16 operands on the matrix cores, f32 storage + accumulate (reduced precision: not the headline)",
               "data": "synthetic",
               "config": {"workload": "Accel-%s (R101-DCN key branch + FlowNet-S warp + R%s correction branch + fused score tail), "
                                      "%dx%d clips, key-frame interval %d, %d clip(s) (1 key + %d non-key frames each) per GPU per step, "
                                      "processed %d clips at a time (batched frames of independent clips)"
                                      % (a.version, a.version, H, W, a.interval, B, a.interval - 1, B),
                          "frames_per_step_per_gpu": a.interval * B, "clips_per_call": B, "root_relief_clips": relief,
                          "inputs": ("frames resident in HBM, COPIED into the model's input buffers every frame (the reference executor's "
                                     "_load_general copy; device to device)" if not wl.bind_inputs else
                                     "frames resident in HBM, read where they lie (--bind-inputs: zero-copy, not the reference's semantics)"),
                          "parallelism": "clip-sharded x%d (weights replicated)" % world,
                          "gather": gather_note + ("; DISABLED after failure: " + gather_failures[0] if gather_failures else ""), "weights": "seeded random",
                          "lowering": ("exact linear folds on (DESIGN.md 4): feat_upsampling*fc6 composed into one deconvolution; non-key L-head fc6 "
                                       "taken from the warped W_fc6*feat image of the key frame" if os.environ.get("ACCEL_FOLD_LINEAR", "1") != "0"
                                       else "reference layer list one to one (ACCEL_FOLD_LINEAR=0)"), "outputs": "fp32 logits 19xHxW + uint8 labels, left in HBM"}}
    if rank == 0 and gather_detail is not None:
        out["gather_rates"] = gather_detail
    if rank == 0 and preflight is not None:
        out["preflight"] = preflight
    if wd is not None:
        wd.close()
    if rank == 0 and rank_ms is not None:
        out["rank_ms_per_step"] = rank_ms
    if rank == 0 and not a.no_roofline:
        out["roofline"] = pair_traffic(wl.conv_roofline(a.dtype), *_pmc_traffic(a.version, H, W, a.interval, a.dtype, B))

    # ---- secondary measurements, same harness (single GPU, headline configuration only) ----------------------------------
    if rank == 0 and world == 1 and dist is None and a.secondary == "auto" and headline_cfg:
        sec = {}
        steps2, warm2 = max(2, a.steps // 2), 1

        def rate(w, el, steps):
            return round(steps * a.interval * w.B / el, 2)
        try:        # the headline loop with the frames read where they lie instead of copied (accel_model_bind_device)
            wl.bind_inputs = True
            el = wl.timed(steps2, warm2)
            sec["accel18_batch%d_zero_copy_inputs" % B] = {
                "value": rate(wl, el, steps2), "unit": "frames/s",
                "what": "the headline workload with the resident frames bound (zero-copy) instead of copied into the model's input buffers: "
                        "9 device-to-device copies of %d MB per step less; an extension, not the reference executor's semantics" % (wl.nbytes >> 20)}
        except Exception as e:
            sec["accel18_batch%d_zero_copy_inputs" % B] = {"error": repr(e)}
        finally:
            wl.bind_inputs = a.bind_inputs
        sec.update(_gather_self_secondary(wl, a, B, H, W, local_rank, steps2, warm2, rate))
        try:
            el = wl.timed_pcie(steps2, warm2)
            sec["accel18_batch%d_pcie_inclusive" % B] = {
                "value": rate(wl, el, steps2), "unit": "frames/s",
                "what": "same %d clips per call; per frame: image from page-locked host memory (next frame's upload overlaps the "
                        "running forward), forward, uint8 label map back on the host" % B}
        except Exception as e:
            sec["accel18_batch%d_pcie_inclusive" % B] = {"error": repr(e)}
        wl.close()
        for name, version, b in (("accel18_batch1", "18", 1), ("accel101_batch%d" % B, "101", B)):
            try:
                w2 = Workload(version, b, H, W, a.interval, local_rank, rank, config)
                el = w2.timed(steps2 * (4 if b == 1 else 1), warm2)
                rf = w2.conv_roofline(a.dtype)
                sec[name] = {"value": rate(w2, el, steps2 * (4 if b == 1 else 1)), "unit": "frames/s", "clips_per_call": b,
                             "what": "Accel-%s, %d clip(s) per call, resident in HBM (the headline's definition)%s"
                                     % (version, b, "; the reference's TEST.BATCH_IMAGES: 1" if b == 1 else ""),
                             "conv_algorithmic_tflops": rf["all_conv"]["algorithmic_tflops"], "conv_executed_frac_of_peak": rf["all_conv"]["frac"]}
                if b == 1:
                    el = w2.timed_pcie(steps2 * 4, warm2)
                    v = rate(w2, el, steps2 * 4)
                    sec["accel18_batch1_pcie_inclusive"] = {
                        "value": v, "unit": "frames/s",
                        "what": "the reference's timing definition (demo.py:234-250) at its own batch: host frame in, forward, label "
                                "map on the host, per frame; page-locked frames, next upload overlapped"}
                    out["vs_baseline"] = round(v / K80_ACCEL18_FPS, 2)
                w2.close()
            except Exception as e:
                sec[name] = {"error": repr(e)}
        if a.dtype == "f32" and "split" not in os.environ.get("ACCEL_WITHHOLD", "").split(","):
            # the headline again with the fp32 MFMA for EVERY product (no bf16x3 launch geometries): the A/B of that choice
            try:
                os.environ["ACCEL_WITHHOLD"] = "split"
                w3 = Workload(a.version, B, H, W, a.interval, local_rank, rank, config)
                el = w3.timed(steps2, warm2)
                rf = w3.conv_roofline(a.dtype)
                sec["accel18_batch%d_fp32_mfma_only" % B] = {
                    "value": rate(w3, el, steps2), "unit": "frames/s", "clips_per_call": B,
                    "what": "the headline workload with ACCEL_WITHHOLD=split: every multiply-add on v_mfma_f32_* (Winograd where it wins), "
                            "no 3 x bf16 split launch geometries",
                    "conv_algorithmic_tflops": rf["all_conv"]["algorithmic_tflops"], "conv_executed_frac_of_peak": rf["all_conv"]["frac"]}
                w3.close()
            except Exception as e:
                sec["accel18_batch%d_fp32_mfma_only" % B] = {"error": repr(e)}
            finally:
                os.environ.pop("ACCEL_WITHHOLD", None)
        if a.dtype == "f32" and "split" not in os.environ.get("ACCEL_WITHHOLD", "").split(",") and os.environ.get("ACCEL_SPLIT", "h2") == "h2":
            # the headline again with the range-free three-term bf16 split (six products) instead of fp16x2 (three): the A/B of the form
            try:
                os.environ["ACCEL_SPLIT"] = "b3"
                w4 = Workload(a.version, B, H, W, a.interval, local_rank, rank, config)
                el = w4.timed(steps2, warm2)
                rf = w4.conv_roofline(a.dtype)
                sec["accel18_batch%d_bf16x3_split" % B] = {
                    "value": rate(w4, el, steps2), "unit": "frames/s", "clips_per_call": B,
                    "what": "the headline workload with ACCEL_SPLIT=b3: every matrix-core geometry in its bf16x3 form (three exact bf16 "
                            "terms per operand, six products, no range slots) instead of fp16x2",
                    "conv_algorithmic_tflops": rf["all_conv"]["algorithmic_tflops"], "conv_executed_frac_of_peak": rf["all_conv"]["frac"]}
                w4.close()
            except Exception as e:
                sec["accel18_batch%d_bf16x3_split" % B] = {"error": repr(e)}
            finally:
                os.environ.pop("ACCEL_SPLIT", None)
        # BASELINE config 5 (reduced precision, never the headline): Accel-50, fp16-MFMA convolutions (operands rounded to half
        # by the loader, fp32 storage + accumulate), 2048x4096, key-frame interval 10, one clip
        try:
            os.environ["ACCEL_CONV_DTYPE"] = "f16"
            w5 = Workload("50", 1, 2048, 4096, 10, local_rank, rank, config)
            el = w5.timed(2, 1)
            rf = w5.conv_roofline("f16")
            sec["accel50_f16_2048x4096_kf10"] = {
                "value": round(2 * 10 / el, 2), "unit": "frames/s", "clips_per_call": 1,
                "what": "BASELINE config 5 on one GPU: Accel-50, fp16-MFMA convolutions (operands rounded to half, ONE product, fp32 accumulation) with "
                        "HALF activation storage between half-capable layers (REDUCED precision: error against the fp32 oracle 4-5 % of the logit range "
                        "at the worst pixel, tests/test_f16_gpu.py, tests/test_f16_storage_gpu.py), 2048x4096, kf=10",
                "conv_algorithmic_tflops": rf["all_conv"]["algorithmic_tflops"], "conv_executed_frac_of_fp16_peak": rf["all_conv"]["frac"]}
            w5.close()
        except Exception as e:
            sec["accel50_f16_2048x4096_kf10"] = {"error": repr(e)}
        finally:
            os.environ["ACCEL_CONV_DTYPE"] = a.dtype
        # ... and the same model, size and schedule in the DEFAULT arithmetic (fp32-class accuracy, fp16x2 form): what the reduced precision buys
        try:
            w6 = Workload("50", 1, 2048, 4096, 10, local_rank, rank, config)
            el = w6.timed(2, 1)
            rf = w6.conv_roofline(a.dtype)
            sec["accel50_2048x4096_kf10_fp16x2"] = {
                "value": round(2 * 10 / el, 2), "unit": "frames/s", "clips_per_call": 1,
                "what": "Accel-50, 2048x4096, kf=10, one clip, in the default arithmetic of the headline (fp32 storage and accuracy, fp16x2 form of the "
                        "matrix-core layers): the line config 5's f16 mode has to be read against",
                "conv_algorithmic_tflops": rf["all_conv"]["algorithmic_tflops"], "conv_executed_frac_of_peak": rf["all_conv"]["frac"]}
            w6.close()
        except Exception as e:
            sec["accel50_2048x4096_kf10_fp16x2"] = {"error": repr(e)}
        out["secondary"] = sec
    else:
        wl.close()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.version, a.interval, H, W)

    def finish():
        if dist is not None:
            with _StdoutToStderr():
                dist.barrier()
                dist.destroy_process_group()
    return (out if rank == 0 else None), finish


if __name__ == "__main__":
    main()
