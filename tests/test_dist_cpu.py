"""Multi-GPU path on CPU: clip sharding and the per-frame gather to rank 0 with
the gloo backend, world size 2 (the RCCL path is the same code with CUDA
tensors and the compute stream as current stream)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from accel_amd import dist as adist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_clips_partitions_exactly():
    for n, w in [(64, 8), (10, 4), (3, 8), (0, 2), (7, 1)]:
        parts = [adist.shard_clips(n, w, r) for r in range(w)]
        flat = [c for p in parts for c in p]
        assert flat == list(range(n))                       # disjoint, ordered, complete
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= (n + w - 1) // w
    assert adist.shard_clips(64, 8, 3) == list(range(24, 32))   # config 4: 8 contiguous clips per GPU


def test_shard_clips_with_a_relieved_root():
    """the root of the logits gather also receives every other rank's frames: root_relief takes clips off rank 0 and hands them to
    the least loaded peers; the partition stays contiguous, disjoint and complete"""
    for n, w, r in [(64, 8, 1), (64, 8, 2), (10, 4, 1), (3, 8, 1), (16, 2, 3), (7, 1, 2), (0, 4, 1)]:
        parts = [adist.shard_clips(n, w, k, root_relief=r) for k in range(w)]
        assert [c for p in parts for c in p] == list(range(n)), (n, w, r)
        even = [len(adist.shard_clips(n, w, k)) for k in range(w)]
        if w > 1:
            base0 = n // w + (1 if n % w else 0)
            assert len(parts[0]) == max(0, base0 - r) and len(parts[0]) <= even[0]
            assert sum(len(p) for p in parts[1:]) == n - len(parts[0])
            assert max(len(p) for p in parts[1:]) - min(len(p) for p in parts[1:]) <= 1 + (1 if n % w else 0)
    assert [len(adist.shard_clips(64, 8, k, root_relief=1)) for k in range(8)] == [7, 8, 8, 8, 8, 8, 8, 9]
    assert [len(adist.shard_clips(64, 8, k, root_relief=0)) for k in range(8)] == [8] * 8
    assert adist.shard_clips(7, 1, 0, root_relief=2) == list(range(7))          # nobody to relieve


def _comm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if rank == 1:
        os.environ["ACCEL_RCCL_UNAVAILABLE"] = "1"       # this rank behaves as if librccl could not be loaded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from accel_amd import runtime

        class Ctx(object):
            device_id = 0
        try:
            adist.make_comm(Ctx())
            q.put((rank, "created"))
        except runtime.AccelError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_no_rank_enters_the_communicator_init_when_one_rank_cannot_load_rccl():
    """ncclCommInitRank blocks until every rank has arrived, so a rank that cannot resolve librccl must be found out BEFORE anybody
    calls it: every rank probes (accel_comm_available) and the ranks vote first.  Rank 1 is made unable here; both ranks must come
    back with AccelError (and FrameGather would fall back to torch.distributed on both) instead of rank 0 hanging in the init."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert "ACCEL_RCCL_UNAVAILABLE" in res[1] and "rank 1" in res[1]
    assert "another rank cannot resolve librccl" in res[0] or "librccl" in res[0]
    assert "created" not in res.values()


def test_greedy_video_assignment_matches_reference_rule():
    # dff_rfcn/function/test_rcnn.py:62-68: next video -> device with the fewest frames so far
    owner, loads = adist.assign_videos_greedy([30, 10, 20, 5, 5, 40], 3)
    assert owner == [0, 1, 2, 1, 1, 1] or owner == [0, 1, 2, 1, 1, 2]
    assert sum(loads) == 110 and max(loads) - min(loads) <= 40


class _FakeModel(object):
    """stands in for runtime.Model: one persistent output buffer per rank"""

    def __init__(self, rank):
        self.rank, self.frame = rank, 0

    def produce(self, shape):
        self.frame += 1
        self.cur = np.full(shape, 100.0 * self.rank + self.frame, np.float32)

    def read(self, name, shape, dtype):
        return self.cur.reshape(shape).astype(dtype)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (19, 8, 16)
        model = _FakeModel(rank)
        g = adist.FrameGather(model, None, "logits", shape, "f4", 0, backend_device="cpu")
        seen = []
        for t in range(5):                               # 5 frames: both staging slots get reused
            model.produce(shape)
            slot = g.submit()
            g.work[slot].wait()
            if rank == 0:
                seen.append([float(x[0, 0, 0]) for x in g.last(slot)])
        g.drain()
        clips = adist.shard_clips(6, world, rank)
        q.put((rank, seen, clips))
    finally:
        dist.destroy_process_group()


def test_gather_logits_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, seen, clips = q.get(timeout=120)
        res[r] = (seen, clips)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen0 = res[0][0]
    assert seen0 == [[1.0 + t, 101.0 + t] for t in range(5)]     # rank order preserved, frame t from every rank
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4, 5]


class _FakeScoreModel(object):
    """stands in for runtime.Model: a `scores` buffer per rank and the expansion rule (here: every score repeated 2 x 2)"""

    def __init__(self, rank):
        self.rank, self.frame = rank, 0

    def produce(self, shape):
        self.frame += 1
        self.cur = (np.arange(np.prod(shape), dtype=np.float32).reshape(shape) % 7) + 100.0 * self.rank + self.frame

    def read(self, name, shape, dtype):
        assert name == "scores"
        return self.cur.reshape(shape).astype(dtype)

    @staticmethod
    def expand_scores_host(maps):
        return np.repeat(np.repeat(maps[..., :19], 2, axis=1), 2, axis=2)


def _score_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, H, W = 2, 64, 96
        model = _FakeScoreModel(rank)
        g = adist.ScoreGather(model, None, B, H, W, 0, backend_device="cpu")
        ok = True
        for t in range(5):                               # both staging slots get reused
            model.produce((B, H // 16, W // 16, 20))
            slot = g.submit()
            if rank == 0:
                exp = g.expanded(slot)
                for r in range(world):
                    want = (np.arange(B * 4 * 6 * 20, dtype=np.float32).reshape(B, 4, 6, 20) % 7) + 100.0 * r + (t + 1)
                    ok = ok and np.array_equal(exp[r], _FakeScoreModel.expand_scores_host(want)) and exp[r].shape == (B, 8, 12, 19)
        g.drain()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gather_scores_world2_gloo():
    """dist.ScoreGather (the default payload of the N > 1 bench): every rank's fused score maps arrive on rank 0 in rank order, frame by
    frame, and are expanded there by the model's own rule"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_score_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment must start the two ranks itself (under
    torch.distributed.run on 127.0.0.1) -- never run one rank and report it as two.  --launch-check stops every rank
    before any GPU work, so the launch path is testable here."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in recs) == [0, 1] and all(r["world"] == 2 and r["self_launched"] for r in recs)
    # a launcher that started another number of ranks than --gpus asks for is refused
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--launch-check"], env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "refusing" in (bad.stderr + bad.stdout)


def _watchdog_worker(rank, world, port, path):
    """rank 1 never reaches the phase rank 0 waits in; rank 0's watchdog must end rank 0 and name rank 1"""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = open(path % rank, "w")
    wd = adist.Watchdog(rank, world, adist.default_store(), stream=log)
    with wd.phase("communicator", 30):
        dist.barrier()              # both arrive: the phase closes, nobody is ended
    if rank == 1:
        time.sleep(20)              # (never enters the next phase while rank 0 waits; ended by the parent)
        return
    with wd.phase("preflight gather", 1.5):
        time.sleep(60)              # stands for an ncclRecv the peer never matches: cannot be interrupted, only the process can go


def test_watchdog_ends_a_rank_whose_peer_never_arrives_and_names_it(tmp_path):
    """Round-5 review, item 8: the first multi-GPU attempt must fail fast and say which rank did not arrive (the C-ABI gathers have no
    timeout).  World size 2 on gloo: both ranks pass the first phase; rank 1 then stays away from the second, rank 0's watchdog
    ends rank 0 with exit code 3 within its deadline and the message names rank 1 as the one that never reached the phase."""
    import time
    ctx = mp.get_context("spawn")
    port = _free_port()
    path = str(tmp_path / "wd%d.log")
    procs = [ctx.Process(target=_watchdog_worker, args=(r, 2, port, path)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    procs[0].join(60)
    took = time.time() - t0
    procs[1].terminate()
    procs[1].join(30)
    assert procs[0].exitcode == adist.Watchdog.EXIT_CODE, procs[0].exitcode
    assert took < 45, "rank 0 was not ended by its 1.5 s deadline (%.0f s)" % took
    text = open(path % 0).read()
    assert "phase 'preflight gather' made no progress" in text and "never reached it: [1]" in text and "still inside it: [0]" in text, text
    assert "communicator" not in text


def test_watchdog_beat_keeps_a_long_loop_alive_and_close_disarms():
    import time
    ended = []
    wd = adist.Watchdog(0, 1, None, _exit=lambda code: ended.append(code))
    with wd.phase("timed steps", 1.0):
        for _ in range(8):
            time.sleep(0.3)
            wd.beat()               # 2.4 s of steps under a 1 s per-step deadline
    time.sleep(0.6)
    assert ended == []
    wd.close()
