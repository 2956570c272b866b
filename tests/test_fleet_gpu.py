"""Fleet readiness that can be checked on ONE GPU (one independent map per agent, no collective): four
maps driven (a) by four processes and (b) by four threads of one process, all on device 0.  Per-map results must
equal the single-map run, and the threads must not serialise each other behind process-wide state (the staging
locks are per map, nothing in the library is global): aggregate throughput of (b) >= 0.8 x (a).  Also runs the
world-size-2 rendezvous of bench.py with REAL plan cycles."""
import os
import socket
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

MAP = "G200"  # 200 x 200 x 50 voxels per agent: four of them keep one GPU busy for the cycles timed here
N_AGENTS = 4
CYCLES = 150


def _build(seed):
    import bench
    map_size, box, occ, ctrl, _ = bench.build_inputs(MAP, seed=seed, n_traj=16)
    return bench.GpuCycle(map_size, box, occ, ctrl, device=0)


def _close(cyc):
    cyc.dev_problem.close()
    cyc.ff.close()  # the finder and the batch hold the map: they go first
    cyc.map.close()


def _fingerprint(cyc):
    """what one plan cycle leaves behind: cluster cell lists, a strided ESDF sample, the batch costs"""
    cl = [c.copy() for c in cyc.ff.clusters(0)]
    d = cyc.map.syncHost(distance=True)["distance"][::211].copy()
    cost, grad = cyc.dev_problem.download()
    return cl, d, cost.copy()


def _same(a, b):
    return (len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and
            np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]))


def _agent_process(seed, q, go):
    cyc = _build(seed)
    cyc.run_native(20)
    cyc.finish()
    q.put(("ready", seed))
    go.wait()
    t0 = time.perf_counter()
    cyc.run_native(CYCLES)
    cyc.finish()
    dt = time.perf_counter() - t0
    cl, d, cost = _fingerprint(cyc)
    q.put(("done", seed, dt, [c.tolist() for c in cl], d.tolist(), cost.tolist()))


def test_four_maps_as_processes_and_as_threads_on_one_device():
    import torch.multiprocessing as mp
    seeds = [42 + k for k in range(N_AGENTS)]
    # reference: every map alone
    alone = {}
    for s in seeds:
        cyc = _build(s)
        cyc.run_native(3)
        cyc.finish()
        alone[s] = _fingerprint(cyc)
        assert len(alone[s][0]) > 0
        _close(cyc)
    # (b) four threads of this process, each with its own map / finder / batch (ctypes drops the GIL in the calls)
    cycs = [_build(s) for s in seeds]
    for c in cycs:
        c.run_native(20)
        c.finish()
    start = threading.Barrier(N_AGENTS + 1)
    errs = []

    def run(c):
        try:
            start.wait()
            c.run_native(CYCLES)
            c.finish()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(c,)) for c in cycs]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt_threads = time.perf_counter() - t0
    assert not errs, errs
    for s, c in zip(seeds, cycs):
        assert _same(_fingerprint(c), alone[s]), "map %d differs when four maps share the process" % s
    for c in cycs:
        _close(c)
    # (a) four processes
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    go = ctx.Event()
    procs = [ctx.Process(target=_agent_process, args=(s, q, go)) for s in seeds]
    for p in procs:
        p.start()
    for _ in range(N_AGENTS):
        assert q.get(timeout=300)[0] == "ready"
    t0 = time.perf_counter()
    go.set()
    res = [q.get(timeout=300) for _ in range(N_AGENTS)]
    dt_procs = time.perf_counter() - t0
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res:
        _, s, dt, cl, d, cost = r
        got = ([np.array(c, dtype=np.int32) for c in cl], np.array(d), np.array(cost))
        assert _same(got, alone[s]), "map %d differs when four processes share the device" % s
    thr_threads = N_AGENTS * CYCLES / dt_threads
    thr_procs = N_AGENTS * CYCLES / dt_procs
    print("fleet on one device: %d threads %.0f cycles/s, %d processes %.0f cycles/s" %
          (N_AGENTS, thr_threads, N_AGENTS, thr_procs))
    # (a wall-clock ratio: printed here, asserted only under -m perf -- tests/test_perf_gpu.py)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    cyc = _build(42 + rank)
    cyc.run_native(5)
    elapsed = bench.timed_fleet_run(lambda: cyc.run_native(40), cyc.finish, 1, dist, None, device="cpu")
    q.put((rank, elapsed, cyc.n_clusters, bench.fleet_value(world, 40, elapsed)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rendezvous_with_real_plan_cycles():
    """bench.py's N > 1 plumbing (barrier, MAX over ranks, whole-job aggregate) around the REAL GpuCycle: two ranks,
    both on the one visible device (the driver's 8-GPU run gives each rank its own)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, e0, n0, v0), (r1, e1, n1, v1) = res
    assert abs(e0 - e1) < 1e-9 and e0 > 0
    assert n0 > 0 and n1 > 0
    assert abs(v0 - 2 * 40 / e0) < 1e-6 * v0


def test_map_destroyed_before_its_finder_and_batch():
    """garbage collectors and destructor orders tear a map down before the objects created on it: their own
    destroy calls must not touch the freed map"""
    cyc = _build(42)
    cyc.run_native(2)
    cyc.finish()
    cyc.map.close()           # first the map ...
    cyc.ff.close()            # ... then its finder
    cyc.dev_problem.close()   # ... and the B-spline batch


def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` on its own (no torchrun around it) launches two ranks and prints ONE line with
    n_gpus = 2.  On a one-GPU box the ranks share the device (FUELMI_FLEET_SHARE_DEVICE=1, rendezvous over gloo);
    the aggregate must be in the range of what two maps sharing a device reach as threads."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FUELMI_FLEET_SHARE_DEVICE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--workload", MAP, "--cpu-budget", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["scaling"] == "weak"
    # the N > 1 line carries the fleet-vs-fleet CPU baseline: one single-threaded process per rank, pinned, summed
    cb = out["cpu_baseline"]
    assert cb["cores"] == 2 and len(cb["per_process"]) == 2 and abs(cb["value"] - sum(cb["per_process"])) < 1e-3
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--workload", MAP,
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert single["n_gpus"] == 1
    # two ranks on ONE device: the whole-job figure is the sum over ranks; what matters is that it IS a two-rank
    # aggregate (n_gpus, one line) and a positive rate -- the range it lands in is a -m perf statement
    assert out["value"] > 0 and single["value"] > 0
    print("bench --gpus 2 on one device: %.0f cycles/s vs %.0f for one rank" % (out["value"], single["value"]))
