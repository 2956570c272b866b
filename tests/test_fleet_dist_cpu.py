"""world_size-2 (gloo, CPU) test of the N>1 path of bench.py: ranks are independent map replicas,
so the only distributed logic is the barrier-bracketed timing, the MAX over ranks and the whole-job
aggregation.  The GPU cycle itself is replaced by a stub here -- this test is about the plumbing."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = {"step": 0, "finish": 0}
    per_step = 0.02 if rank == 0 else 0.05  # rank 1 is the slow agent

    def step():
        calls["step"] += 1
        time.sleep(per_step)

    def finish():
        calls["finish"] += 1

    elapsed = bench.timed_fleet_run(step, finish, 5, dist, None, device="cpu")
    q.put((rank, elapsed, calls["step"], calls["finish"], bench.fleet_value(world, 5, elapsed)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fleet_timing_is_max_over_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, s0, f0, v0), (r1, e1, s1, f1, v1) = res
    assert s0 == s1 == 5 and f0 == f1 == 2          # exactly K steps, sync on both sides
    assert abs(e0 - e1) < 1e-9                      # both ranks report the MAX
    assert 0.25 <= e0 < 0.6                         # >= 5 * 0.05 s of the slow rank
    assert abs(v0 - 2 * 5 / e0) < 1e-9              # whole-job aggregate, not per-GPU


def test_workload_table_matches_baseline_configs():
    import bench
    assert bench.WORKLOADS["G400"][0] == (40.0, 40.0, 10.0)
    assert bench.WORKLOADS["G800"][0] == (80.0, 80.0, 20.0)
    lo, hi = bench.exploration_box((40.0, 40.0, 10.0))
    assert lo == (-19.0, -19.0, 0.0) and hi == (19.0, 19.0, 7.0)


def test_bench_gpus_flag_refuses_to_run_one_gpu_silently():
    """`python bench.py --gpus N` launches its own N ranks; with fewer devices than ranks (none here) it must say so
    instead of quietly measuring one GPU (--gpus was parsed and ignored)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FUELMI_FLEET_SHARE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout) and "device" in (r.stderr + r.stdout)


def _baseline_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    mine = {"value": 2.0 + rank, "unit": "cycles/s", "cores": 1, "kind": "reference", "sample": "stub of rank %d" % rank}
    out = bench.fleet_cpu_baseline(mine, rank, world, dist, core=10 + rank)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_fleet_cpu_baseline_is_the_sum_of_one_pinned_process_per_rank():
    """N > 1 bench lines carry a cpu_baseline too: N single-threaded reference processes, one per
    core, run concurrently; rank 0 reports their summed rate with cores = N."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_baseline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    b = res[0]
    assert b["value"] == 5.0 and b["cores"] == 2 and b["kind"] == "reference"
    assert b["per_process"] == [2.0, 3.0] and b["pinned_cores"] == [10, 11]
