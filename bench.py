#!/usr/bin/env python3
"""bench.py -- plan cycles/s of the MI355X hot path (BASELINE.json metric).

One "step" = one full plan cycle on one 400x400x100 @ 0.1 m map (BASELINE configs[1], with the
64-candidate B-spline batch the metric is quoted on), inputs already resident in HBM:
    clearAndInflateLocalMap (full box)  ->  updateESDF3d (full box)
    -> FrontierFinder::searchFrontiers over the whole exploration box (fresh flags)
    -> 64 x combineCost (NORMAL_PHASE|MINTIME, 32 control points each)
Every rank owns an independent map on its own GPU (RACER-style fleet; the path has no exchange
step, so there is NO data-path collective -- "weak" scaling by construction).  torch is used only
for the process group (barrier / max-reduce of the timing) and device selection.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (map_size m, n_obstacles, n_known_spheres)
    "G400": ((40.0, 40.0, 10.0), 400, 120),
    "G800": ((80.0, 80.0, 20.0), 3200, 960),
    "G100": ((10.0, 10.0, 5.0), 25, 8),  # smoke-sized
    "G200": ((20.0, 20.0, 5.0), 100, 30),  # fleet test: four of these share one GPU
    # ESDF worst-case regimes: the whole map explored, optimistic = true (only inflated
    # obstacle voxels are sources, topo_algorithm.xml:71-72): distances of 10-100 voxels instead of ~1.4
    "G400K": ((40.0, 40.0, 10.0), 400, -1),   # the headline world, fully known
    "G400E": ((40.0, 40.0, 10.0), 12, -1),    # a nearly empty hall: floor + a dozen obstacles
    # streaming variants (BASELINE config #4): the map starts unknown, one 640x480 depth frame per step
    # (sparser worlds than the full-box recipe, so that a frame sees several metres of free space)
    "G800S": ((80.0, 80.0, 20.0), 600, 0),
    "G400S": ((40.0, 40.0, 10.0), 150, 0),
}


def exploration_box(map_size):
    org = (-map_size[0] / 2.0, -map_size[1] / 2.0, -1.0)
    lo = (org[0] + 1.0, org[1] + 1.0, 0.0)
    hi = (-org[0] - 1.0, -org[1] - 1.0, max(0.8 * map_size[2] - 1.0, 1.0))
    return lo, hi


def make_trajectories(rng, n_traj, n_pts, lo, hi, seg_len=6.0, noise=0.3):
    ctrl = np.empty((n_traj, n_pts, 3))
    lo = np.asarray(lo, dtype=float)
    hi = np.asarray(hi, dtype=float)
    for c in range(n_traj):
        a = lo + (hi - lo) * rng.random(3)
        d = rng.normal(size=3)
        d[2] *= 0.2
        d /= np.linalg.norm(d)
        b = np.clip(a + seg_len * d, lo, hi)
        t = np.linspace(0, 1, n_pts)[:, None]
        ctrl[c] = a + (b - a) * t + rng.normal(scale=noise, size=(n_pts, 3))
    return ctrl


def bspline_problem(ctrl, dt):
    """NLopt-layout variables + boundary states (host arrays) for a batch of candidates."""
    C, N, _ = ctrl.shape
    x = np.concatenate([ctrl.reshape(C, N * 3), np.full((C, 1), dt)], axis=1)
    seg = np.linalg.norm(np.diff(ctrl, axis=1), axis=2).sum(axis=1)
    pt_dist = seg / float(N)  # optimize(): sum |dq| / point_num (bspline_optimizer.cpp:136-140)
    start = np.zeros((C, 3, 3))
    end = np.zeros((C, 3, 3))
    start[:, 0] = (ctrl[:, 0] + 4 * ctrl[:, 1] + ctrl[:, 2]) / 6.0
    start[:, 1] = (ctrl[:, 2] - ctrl[:, 0]) / (2 * dt)
    end[:, 0] = (ctrl[:, -1] + 4 * ctrl[:, -2] + ctrl[:, -3]) / 6.0
    return np.ascontiguousarray(x), pt_dist, start, end


def build_inputs(workload, seed, n_traj=64):
    """Synthetic map state (host) for one agent: occupancy log-odds + candidate trajectories."""
    from fuel_amd import synth
    map_size, n_obs, n_sph = WORKLOADS[workload]
    w = synth.World.for_map_size(map_size)
    truth = w.world(seed, n_obs)
    if n_sph < 0:  # everything explored: free -> clamp_min_log, solid -> clamp_max_log
        lo5 = synth.logodds()
        occ = np.where(truth.astype(bool), lo5[3], lo5[2]).astype(np.float64)
        n_known = occ.size
    else:
        occ, n_known = w.known_state(truth, seed, n_sph)
    lo, hi = exploration_box(map_size)
    rng = np.random.default_rng(1000 + seed)
    ctrl = make_trajectories(rng, n_traj, 32, np.array(lo) + 0.5, np.array(hi) - 0.5)
    return map_size, (lo, hi), occ, ctrl, n_known


class GpuCycle:
    """The hot path on one GPU through the C-ABI (fuel_amd.host mirrors the reference classes)."""

    def __init__(self, map_size, box, occ, ctrl, device, dt=0.175, reference_order=0, **map_kw):
        import fuel_amd
        self.fa = fuel_amd
        self.map = fuel_amd.SDFMap(map_size, box[0], box[1], device=device, **map_kw)
        self.map.uploadOccupancy(occ)
        nv = self.map.nvox
        self.map.setLocalBound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
        self.box = box
        self.ff = fuel_amd.FrontierFinder(self.map, cluster_min=100, reference_order=reference_order)
        self.opt = fuel_amd.BsplineOptimizer()
        self.opt.setEnvironment(self.map)
        x, ptd, st, en = bspline_problem(ctrl, dt)
        cf = fuel_amd.NORMAL_PHASE | fuel_amd.MINTIME
        self.problem = fuel_amd.BsplineBatchProblem(x, ctrl.shape[1], cf, ptd, st, en, 3, 3, dt)
        self.dev_problem = self.opt.deviceProblem(self.problem)
        self.n_clusters = 0

    def step_serial(self):
        """Diagnostic order (--serial-stages): no overlap between the ESDF chain and the frontier scan."""
        m = self.map
        m.clearAndInflateLocalMap()
        m.updateESDF3d()
        self.dev_problem.eval()
        m.synchronize()
        self.ff.reset()
        m.setUpdatedBox(self.box[0], self.box[1])
        self.n_clusters = self.ff.searchFrontiers()

    def step(self):
        m = self.map
        # the frontier scan reads only the occupancy state: enqueue it first (own stream), then the
        # inflation -> ESDF -> B-spline chain on the map's stream; collect the clusters last
        self.ff.reset()
        m.setUpdatedBox(self.box[0], self.box[1])
        self.ff.searchFrontiersBegin()
        m.clearAndInflateLocalMap()
        m.updateESDF3d()
        self.dev_problem.eval()
        self.n_clusters = self.ff.searchFrontiersEnd()

    def finish(self):
        self.map.synchronize()
        self.ff.sync()

    def run_native(self, n, serial=False):
        """n cycles of step() / step_serial() issued from C++ (fuelmi_bench_cycles): the same seven calls per
        cycle without the interpreter between them."""
        import ctypes as C
        ncl, sec = C.c_int(), C.c_double()
        lo, hi = (C.c_double * 3)(*self.box[0]), (C.c_double * 3)(*self.box[1])
        from fuel_amd._lib import check
        check(self.map.L.fuelmi_bench_cycles(self.map.h, self.ff.h, self.dev_problem.h, lo, hi, int(n), int(serial),
                                             C.byref(ncl), C.byref(sec)))
        self.n_clusters = ncl.value
        return sec.value

    def host_profile(self):
        """mean host microseconds per cycle inside each C-ABI call of the last run_native (fuelmi_bench_host_profile)"""
        import ctypes as C
        out = (C.c_double * 7)()
        from fuel_amd._lib import check
        check(self.map.L.fuelmi_bench_host_profile(self.map.h, out))
        keys = ("reset_and_box", "search_begin", "inflate_local", "update_esdf", "bspline_eval", "search_end",
                "search_end_polling")
        return {k: round(v, 2) for k, v in zip(keys, out)}


def run_delivered(cyc, n):
    """n cycles with the results delivered to host memory every cycle (fuelmi_bench_cycles_delivered): the cells of
    every new cluster and the cost / gradient of every candidate, as the reference's callers receive them."""
    import ctypes as C
    from fuel_amd._lib import check
    ncl, sec = C.c_int(), (C.c_double * 3)()
    lo, hi = (C.c_double * 3)(*cyc.box[0]), (C.c_double * 3)(*cyc.box[1])
    cap = 1 << 22
    cells = np.empty(cap, dtype=np.int32)
    nc, nvar = cyc.problem.x.shape
    cost, grad = np.empty(nc), np.empty((nc, nvar))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    check(cyc.map.L.fuelmi_bench_cycles_delivered(cyc.map.h, cyc.ff.h, cyc.dev_problem.h, lo, hi, int(n),
                                                  cells.ctypes.data_as(ip), C.c_size_t(cap), cost.ctypes.data_as(dp),
                                                  grad.ctypes.data_as(dp), C.byref(ncl), sec))
    return sec[0], sec[1], sec[2]


def streaming_frames(map_size, n_obs, n_frames, seed):
    """Depth frames (16UC1, 640x480) along a seeded camera path through a fresh synthetic world."""
    from fuel_amd import synth
    w = synth.World.for_map_size(map_size)
    truth = w.world(seed, n_obs)
    frames = []
    n_try = 6 * n_frames
    for k in range(n_try):  # keep the poses of the seeded path that look into open space
        pose = w.camera(truth, 7 + seed, k, n_try, 0.7)
        img = w.depth_image(truth, pose, 640, 480)
        if np.median(np.where(img == 0, 7000, img)) < 2000:  # staring at a wall: skip
            continue
        frames.append((img, pose[:3].copy(), synth.World.pose_quaternion(pose)))
        if len(frames) == n_frames:
            break
    if not frames:
        raise SystemExit("no usable camera pose in the synthetic world")
    return frames


class GpuStreamCycle:
    """BASELINE config #4: streaming depth inserts + incremental (box-local) ESDF + incremental frontier
    search on a map that starts unknown, plus the B-spline batch.  One step = one depth frame."""

    def __init__(self, map_size, box, frames, ctrl, device, dt=0.175, reference_order=0):
        import fuel_amd
        self.map = fuel_amd.SDFMap(map_size, box[0], box[1], device=device)
        self.ff = fuel_amd.FrontierFinder(self.map, cluster_min=100, reference_order=reference_order)
        self.frames = frames
        self.k = 0
        # where the depth frames live when a step hands one over (frame_source):
        #   "device"   -- resident in HBM before the timed region starts (what `value` is quoted on)
        #   "pinned"   -- a registered host ring, read in place over PCIe by the fusion kernels
        #   "pageable" -- an ordinary host array (a cv::Mat): staged through the map's pinned buffer
        stack = np.ascontiguousarray(np.stack([f[0] for f in frames]).astype(np.uint16))
        self.rows, self.cols = stack.shape[1], stack.shape[2]
        self.dev_frames = fuel_amd.DeviceBuffer(stack, device)
        self.pinned = fuel_amd.RegisteredHostBuffer(stack.copy())  # (unregisters itself before its memory is released)
        fb = self.rows * self.cols * 2
        self.ptr = {"device": [self.dev_frames.ptr + i * fb for i in range(len(frames))],
                    "pinned": [self.pinned.ptr + i * fb for i in range(len(frames))]}
        self.frame_source = "device"
        self.opt = fuel_amd.BsplineOptimizer()
        self.opt.setEnvironment(self.map)
        x, ptd, st, en = bspline_problem(ctrl, dt)
        cf = fuel_amd.NORMAL_PHASE | fuel_amd.MINTIME
        self.problem = fuel_amd.BsplineBatchProblem(x, ctrl.shape[1], cf, ptd, st, en, 3, 3, dt)
        self.dev_problem = self.opt.deviceProblem(self.problem)
        self.n_clusters = 0
        self.box_vox = []  # voxels of the local bound of every frame (the ESDF / inflation box)

    def fuse(self, i):
        img, pos, q = self.frames[i]
        if self.frame_source == "pageable":
            return self.map.inputDepthImage(img, pos, q)
        return self.map.inputDepthImageAt(self.ptr[self.frame_source][i], self.rows, self.cols, pos, q)

    def step(self):
        i = self.k % len(self.frames)
        self.k += 1
        m = self.map
        fused = self.fuse(i) > 0                      # MapROS::depthPoseCallback
        # the scan only reads the occupancy planes the fusion just rewrote: queue it first (own stream), the
        # inflation -> ESDF -> B-spline chain beside it, collect the clusters last (same order as GpuCycle)
        self.ff.searchFrontiersBegin()                # consumes the accumulated updated box
        if fused:
            lo, hi = m.getLocalBound()
            self.box_vox.append(float(np.prod(np.array(hi) - np.array(lo) + 1)))
            m.clearAndInflateLocalMap()
            m.updateESDF3d()                          # (the reference defers this to a 50 ms timer)
        self.dev_problem.eval()
        self.n_clusters = self.ff.searchFrontiersEnd()
        self.ff.commit()

    def step_serial(self):
        """Diagnostic order: the frontier search after the map chain instead of beside it."""
        i = self.k % len(self.frames)
        self.k += 1
        m = self.map
        if self.fuse(i) > 0:
            lo, hi = m.getLocalBound()
            self.box_vox.append(float(np.prod(np.array(hi) - np.array(lo) + 1)))
            m.clearAndInflateLocalMap()
            m.updateESDF3d()
        self.dev_problem.eval()
        m.synchronize()
        self.n_clusters = self.ff.searchFrontiers()
        self.ff.commit()

    def run_native(self, n, serial=False):
        return self.prepare_native(n, serial)()

    def prepare_native(self, n, serial=False):
        """the next n frames through step() / step_serial(), issued from C++ (fuelmi_bench_stream): the same calls
        per cycle without the interpreter between them.  Returns the callable that runs them (argument marshalling
        stays outside the timed region)."""
        import ctypes as C
        from fuel_amd._lib import check
        nf = len(self.frames)
        idx = [(self.k + j) % nf for j in range(n)]
        self.k += n
        if self.frame_source == "pageable":
            ptrs = [self.frames[i][0].ctypes.data for i in idx]
        else:
            ptrs = [self.ptr[self.frame_source][i] for i in idx]
        depth = (C.c_void_p * n)(*ptrs)
        pos = np.ascontiguousarray(np.array([self.frames[i][1] for i in idx], dtype=np.float64))
        quat = np.ascontiguousarray(np.array([self.frames[i][2] for i in idx], dtype=np.float64))
        cfg = self.map.depthConfig()
        ncl, vox, sec = C.c_int(), C.c_double(), C.c_double()
        dp = C.POINTER(C.c_double)

        def run():
            check(self.map.L.fuelmi_bench_stream(self.map.h, self.ff.h, self.dev_problem.h, n, depth, self.rows,
                                                 self.cols, C.byref(cfg), pos.ctypes.data_as(dp), quat.ctypes.data_as(dp),
                                                 int(serial), C.byref(ncl), C.byref(vox), C.byref(sec)))
            self.n_clusters = ncl.value
            self.box_vox.extend([vox.value / max(n, 1)] * n)
            return sec.value
        return run

    def finish(self):
        self.map.synchronize()
        self.ff.sync()

    def close(self):
        """drain, release the finder / batch / map and undo the registration of the host ring"""
        self.finish()
        for o in (self.dev_problem, self.ff, self.map):
            o.close()
        if self.pinned is not None:
            self.pinned.close()
            self.pinned = None
        self.dev_frames = None


def cpu_baseline_stream(map_size, box, frames, ctrl, budget_s=12.0, dt=0.175):
    """The CPU oracle on the streaming cycle (projection + fusion + local ESDF + incremental frontiers +
    B-spline batch), bounded sample of consecutive frames."""
    from oracle import fuel_oracle as fo
    om = fo.OracleMap(map_size, box[0], box[1])
    of = fo.OracleFrontier(om, 100)
    x, ptd, st, en = bspline_problem(ctrl, dt)
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    times = []
    t_all = time.perf_counter()
    k = 0
    while True:
        img, pos, q = frames[k % len(frames)]
        k += 1
        t0 = time.perf_counter()
        pts = fo.project_depth(img, pos, q)
        if len(pts):
            om.input_points(pts, pos)
            om.inflate_local()
            om.update_esdf()
        for c in range(ctrl.shape[0]):
            fo.bspline_cost_grad(om, x[c], ctrl.shape[1], cf, ptd[c], st[c], en[c], 3, 3, dt)
        of.search()
        of.commit()
        times.append(time.perf_counter() - t0)
        # one pass over the distinct frames at most: a frame seen again changes nothing and costs far less
        if k >= len(frames) or (len(times) >= 4 and time.perf_counter() - t_all > budget_s):
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "cycles/s", "cores": 1, "kind": "port",
            "sample": "%d consecutive streaming cycles (median), 1 thread, g++ -O3" % len(times)}


def cpu_baseline_stream_reference(map_size, box, frames, ctrl, budget_s=12.0, dt=0.175):
    """The streaming cycle through the reference's own code (oracle/_ref: map_ros.cpp's proessDepthImage in one
    library, sdf_map.cpp / frontier_finder.cpp / bspline_optimizer.cpp in the other), or None."""
    try:
        from oracle.ref_build import ref
        from oracle import fuel_oracle as fo
        if not (ref.available() and ref.mapros_available()):
            return None
    except Exception:
        return None
    rm = ref.RefMap(map_size, box[0], box[1])
    rf = ref.RefFrontier(rm, 100)
    x, ptd, st, en = bspline_problem(ctrl, dt)
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    times = []
    t_all = time.perf_counter()
    k = 0
    while True:
        img, pos, q = frames[k % len(frames)]
        k += 1
        t0 = time.perf_counter()
        pts = ref.project_depth(img, pos, q)
        if len(pts):
            rm.input_points(pts, pos)
            rm.inflate_local()
            rm.update_esdf()
        for c in range(ctrl.shape[0]):
            ref.bspline_cost_grad(rm, x[c], ctrl.shape[1], cf, ptd[c], st[c], en[c], 3, 3, dt)
        rf.search()
        rf.commit()
        times.append(time.perf_counter() - t0)
        # one pass over the distinct frames at most: a frame seen again changes nothing and costs far less
        if k >= len(frames) or (len(times) >= 4 and time.perf_counter() - t_all > budget_s):
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "cycles/s", "cores": 1, "kind": "reference",
            "sample": "%d consecutive streaming cycles (median) through the reference's own map_ros.cpp / sdf_map.cpp / "
                      "frontier_finder.cpp / bspline_optimizer.cpp (oracle/_ref, header stand-ins, g++ -O3, 1 thread)"
                      % len(times)}


def cpu_baseline(map_size, box, occ, ctrl, budget_s=12.0, dt=0.175):
    """The CPU oracle (restatement of the reference, 1 thread) on the same cycle, bounded sample."""
    from oracle import fuel_oracle as fo
    om = fo.OracleMap(map_size, box[0], box[1])
    om.occ[:] = occ
    nv = om.nvox
    om.set_local_bound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
    x, ptd, st, en = bspline_problem(ctrl, dt)
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    times = []
    stage = np.zeros(4)
    t_all = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        om.inflate_local()
        t1 = time.perf_counter()
        om.update_esdf()
        t2 = time.perf_counter()
        of = fo.OracleFrontier(om, 100)
        om.set_updated_box(box[0], box[1])
        of.search()
        t3 = time.perf_counter()
        for c in range(ctrl.shape[0]):
            fo.bspline_cost_grad(om, x[c], ctrl.shape[1], cf, ptd[c], st[c], en[c], 3, 3, dt)
        t4 = time.perf_counter()
        times.append(t4 - t0)
        stage += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        del of
        if len(times) >= 2 and time.perf_counter() - t_all > budget_s:
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "cycles/s", "cores": 1, "kind": "port",
            "sample": "%d full plan cycles of the same G-map (median), 1 thread, g++ -O3; "
                      "stage ms inflate/esdf/frontier/bspline = %s" %
                      (len(times), "/".join("%.1f" % (1e3 * s / len(times)) for s in stage))}


def cpu_baseline_reference(map_size, box, occ, ctrl, budget_s=12.0, dt=0.175):
    """The REAL reference code (oracle/_ref: sdf_map.cpp, frontier_finder.cpp, bspline_optimizer.cpp compiled
    from /root/reference with header stand-ins, prebuilt -- the GPU box only loads the .so) on the same cycle,
    1 thread, bounded sample.  None when the library is not there."""
    try:
        from oracle.ref_build import ref
        from oracle import fuel_oracle as fo
        if not ref.available():
            return None
    except Exception:
        return None
    rm = ref.RefMap(map_size, box[0], box[1])
    rm.occ[:] = occ
    nv = rm.nvox
    rm.set_local_bound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
    x, ptd, st, en = bspline_problem(ctrl, dt)
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    times = []
    stage = np.zeros(4)
    t_all = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        rm.inflate_local()
        t1 = time.perf_counter()
        rm.update_esdf()
        t2 = time.perf_counter()
        rf = ref.RefFrontier(rm, 100)  # cluster_size_xy < 0: searchFrontiers up to (not including) the split
        rm.set_updated_box(box[0], box[1])
        rf.search()
        t3 = time.perf_counter()
        for c in range(ctrl.shape[0]):
            ref.bspline_cost_grad(rm, x[c], ctrl.shape[1], cf, ptd[c], st[c], en[c], 3, 3, dt)
        t4 = time.perf_counter()
        times.append(t4 - t0)
        stage += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        del rf
        if len(times) >= 2 and time.perf_counter() - t_all > budget_s:
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "cycles/s", "cores": 1, "kind": "reference",
            "sample": "%d full plan cycles of the same G-map (median) through the reference's own sdf_map.cpp / "
                      "frontier_finder.cpp / bspline_optimizer.cpp (oracle/_ref, Eigen/ROS/PCL header stand-ins, "
                      "g++ -O3, 1 thread); stage ms inflate/esdf/frontier/bspline = %s" %
                      (len(times), "/".join("%.1f" % (1e3 * s / len(times)) for s in stage))}


def fleet_cpu_baseline(entry, rank, world, dist, core):
    """N > 1: every rank has timed ITS OWN single-threaded CPU baseline on its own map, pinned to its own core, all at
    the same time -- N independent reference processes, one per core, as the reference would run a fleet
    (exploration_node.cpp:19: one ros::spin thread per agent).  Rank 0 gets the aggregate."""
    entries = [None] * world
    dist.all_gather_object(entries, (entry, core))
    if rank != 0:
        return None
    vals = [e[0]["value"] for e in entries]
    e0 = entries[0][0]
    return {"value": float(sum(vals)), "unit": e0["unit"], "cores": world, "kind": e0["kind"],
            "per_process": [round(v, 4) for v in vals], "pinned_cores": [e[1] for e in entries],
            "sample": "%d independent single-threaded processes (one per rank, each pinned to its own core, each on its own "
                      "map, run concurrently); value = sum of their rates.  Each: %s" % (world, e0["sample"])}


def timed_fleet_run(step, finish, steps, dist=None, device_sync=None, device="cuda"):
    """Time exactly `steps` steps, bracketed by barrier + device sync on both sides; returns the MAX
    elapsed seconds over ranks.  No data-path collective: ranks are independent maps."""
    import torch

    def sync_all():
        if dist is not None:
            dist.barrier()
        finish()
        if device_sync is not None:
            device_sync()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def fleet_value(n_ranks, steps, elapsed_max):
    """Whole-job throughput: units all ranks processed / max-over-ranks time."""
    return n_ranks * steps / elapsed_max


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed cycles (default 200 for the full-box workloads: 20 cycles are 4 ms, too short for "
                         "a steady clock; 20 for the streaming ones, one distinct depth frame each)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed cycles first (default 20 / 3)")
    ap.add_argument("--workload", default="G400", choices=sorted(WORKLOADS))
    ap.add_argument("--candidates", type=int, default=64,
                    help="B-spline candidates per cycle (BASELINE configs: 1, 64 = headline, 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--reference-order", type=int, default=0, choices=(0, 1, 2),
                    help="fuelmi_frontier_cfg.reference_order of the finder (0 address order, 1 the reference's BFS order, "
                         "2 auto: the reference's order whenever every cluster of the search holds <= 26624 cells); "
                         "the headline uses 0")
    ap.add_argument("--esdf-family", type=int, default=-1, choices=(-1, 0, 1, 2),
                    help="pin the ESDF kernel family (fuelmi_map_set_esdf_family): -1 the library's per-update choice "
                         "(default), 0 packed plain, 1 far-field, 2 32-bit plain")
    ap.add_argument("--serial-stages", action="store_true",
                    help="diagnostic: run the frontier scan after the ESDF chain instead of beside it")
    args = ap.parse_args()
    stream_wl = args.workload.endswith("S")
    if args.steps is None:
        args.steps = 20 if stream_wl else 200
    if args.warmup is None:
        args.warmup = 3 if stream_wl else 20

    import torch
    share = os.environ.get("FUELMI_FLEET_SHARE_DEVICE") == "1"  # N ranks on however many devices there are (tests)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: launch the N ranks here (one process per GPU, the same command the
        # driver uses) and hand their single JSON line through
        import socket
        import subprocess
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not share:
            raise SystemExit("bench.py --gpus %d: only %d device(s) visible (FUELMI_FLEET_SHARE_DEVICE=1 lets the ranks "
                             "share them)" % (args.gpus, ndev))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the library first: fuel_amd.lib() calls fuelmi_init(), which sets the hardware-queue default (GPU_MAX_HW_QUEUES,
    # unless the environment decides) -- and the HIP runtime reads the variable at its first call, which torch.cuda
    # below would otherwise make
    import fuel_amd as _fa_first
    _fa_first.lib()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr)
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not share:
        raise SystemExit("rank %d has no device (%d visible; FUELMI_FLEET_SHARE_DEVICE=1 lets ranks share)" % (local_rank, ndev))
    local_rank_raw = local_rank
    local_rank = local_rank % ndev  # (the device this rank uses)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:  # RCCL refuses two ranks on one device; the barrier and the max-reduce of the time run over gloo
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world

    # 2 x ranks > cores: the host loops that poll for device results (a core each) would fight the runtime's own
    # threads -- they yield between polls instead (read once by the library)
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    if world > 1 and ncores < 2 * world:
        os.environ.setdefault("FUELMI_POLL_YIELD", "1")
    import fuel_amd
    from fuel_amd import _lib
    if args.esdf_family >= 0:
        fuel_amd.SDFMap.default_esdf_family = args.esdf_family
    streaming = args.workload.endswith("S")
    if streaming:
        map_size, n_obs, _ = WORKLOADS[args.workload]
        box = exploration_box(map_size)
        # distinct frames for every step of the run (warm-up, the two short profiling passes, timed region)
        frames = streaming_frames(map_size, n_obs, args.warmup + 4 * args.steps + 24, seed=42 + rank)
        rng = np.random.default_rng(1000 + 42 + rank)
        ctrl = make_trajectories(rng, args.candidates, 32, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
        occ, n_known = None, 0
        cyc = GpuStreamCycle(map_size, box, frames, ctrl, device=local_rank, reference_order=args.reference_order)
    else:
        map_size, box, occ, ctrl, n_known = build_inputs(args.workload, seed=42 + rank, n_traj=args.candidates)
        map_kw = {"optimistic": 1} if args.workload in ("G400K", "G400E") else {}
        cyc = GpuCycle(map_size, box, occ, ctrl, device=local_rank, reference_order=args.reference_order, **map_kw)
    if args.serial_stages:
        cyc.step = cyc.step_serial

    # W untimed warmup steps, then a short untimed pass with every stage bracketed by HIP events
    # (on the map's own stream) to find the dominant kernel
    stages = {"inflate": _lib.K_INFLATE, "esdf_zy": _lib.K_ESDF_ZY, "esdf_x": _lib.K_ESDF_X,
              "frontier": _lib.K_FRONTIER, "bspline": _lib.K_BSPLINE}
    if streaming:
        stages["insert"] = _lib.K_INSERT  # upload + projection + fusion (+ one host sync): reported, not ranked
    for _ in range(args.warmup):
        cyc.step()
    cyc.finish()
    cyc.map.profileEnable(sum(1 << v for v in stages.values()))
    for _ in range(3):
        cyc.step()
    cyc.finish()
    stage_ms = {}
    for name, sid in stages.items():
        n, tot = cyc.map.profileGet(sid)
        stage_ms[name] = tot / max(n, 1)
    # the device's own timeline of the LAST of those cycles (events, no tracer on the host): begin / end of every stage
    # bracket in microseconds after the first one of that cycle, and the map queue's idle time between its kernels
    timeline = None
    try:
        tl = []
        for name, sid in stages.items():
            a, b = cyc.map.profileTimeline(sid)
            if len(a):
                tl.append((name, 1e3 * float(a[-1]), 1e3 * float(b[-1])))
        if tl:
            t_first = min(t[1] for t in tl)
            tl = sorted((n_, round(a_ - t_first, 1), round(b_ - t_first, 1)) for n_, a_, b_ in tl)
            tl.sort(key=lambda t: t[1])
            mapq = [t for t in tl if t[0] in ("inflate", "esdf_zy", "esdf_x", "bspline", "insert")]
            gaps = {"%s->%s" % (mapq[i][0], mapq[i + 1][0]): round(mapq[i + 1][1] - mapq[i][2], 1) for i in range(len(mapq) - 1)}
            timeline = {"stage_begin_end_us": [list(t) for t in tl], "map_queue_idle_us": gaps,
                        "note": "HIP events of one cycle of the untimed profiling pass (every stage bracketed: the brackets "
                                "themselves add event packets); frontier = the whole chain incl. its tail, on its own stream"}
    except Exception:
        timeline = None
    # dominant kernel = longest single-kernel stage when the two chains do NOT overlap (stable from run
    # to run; inside the overlapped cycle the durations depend on what the other stream happens to run)
    cyc.map.profileEnable(sum(1 << v for v in stages.values()))  # (re-arming clears the event log)
    for _ in range(5):
        cyc.step_serial()
    cyc.finish()
    iso_ms_all = {}
    for name, sid in stages.items():  # median: one disturbed launch must not pick the "dominant" kernel
        smp = cyc.map.profileSamples(sid)
        iso_ms_all[name] = float(np.median(smp)) if len(smp) else 0.0
    kernel_stages = {k: v for k, v in iso_ms_all.items() if k not in ("frontier", "insert")}  # multi-kernel stages
    dominant = max(kernel_stages, key=kernel_stages.get)
    # What a plain streaming kernel reaches on this device (STREAM triad over 3 x 1 GiB), reported beside the vendor
    # peak that `frac` uses.  Measured here, right in front of the timed region: the ~0.1 s of it also leaves the
    # device at its working clocks, which a 20-cycle (2.6 ms) timed region started from idle would not reach.
    triad_gbs = None
    try:
        import ctypes as C
        tri = C.c_double()
        _lib.check(cyc.map.L.fuelmi_hbm_triad(local_rank, 1 << 30, 5, C.byref(tri)))
        triad_gbs = round(tri.value, 1)
    except Exception:
        triad_gbs = None
    expand_gbs = None  # the x pass's read : write mix (1 : 2) as a plain streaming kernel
    try:
        ex = C.c_double()
        _lib.check(cyc.map.L.fuelmi_hbm_expand(local_rank, 1 << 29, 5, C.byref(ex)))
        expand_gbs = round(ex.value, 1)
    except Exception:
        expand_gbs = None
    for _ in range(min(args.warmup, 5)):  # (the state the profiling passes left behind: back to the steady cycle)
        cyc.step()
    cyc.finish()
    # timed region: only the dominant kernel stays bracketed (two event records per step)
    cyc.map.profileEnable(1 << stages[dominant])
    host_loop = None
    if streaming:
        # the K frames are issued by the library's own C++ loop (fuelmi_bench_stream), like the full-box cycles below
        elapsed = timed_fleet_run(cyc.prepare_native(args.steps, args.serial_stages), cyc.finish, 1, dist,
                                  torch.cuda.synchronize)
    else:
        # the K timed cycles are issued by the library's own C++ loop (fuelmi_bench_cycles): the Python
        # interpreter's ~25 us per cycle between the seven C-ABI calls is not part of the hot path.  The
        # interpreter-driven figure is reported beside it (host_loop).
        # A K-cycle region of 20 cycles is 2 ms: one disturbed cycle moves it by 5 %.  Short regions are therefore
        # repeated and the MEDIAN region is the one reported: every repetition times exactly K cycles,
        # bracketed like the contract says; `region_ms` lists them all.
        reps = 9 if args.steps < 100 else 1
        regions = [timed_fleet_run(lambda: cyc.run_native(args.steps, args.serial_stages), cyc.finish, 1, dist,
                                   torch.cuda.synchronize) for _ in range(reps)]
        elapsed = float(np.median(regions))
    n_launch, dom_total_ms = cyc.map.profileGet(stages[dominant])
    host_issue = None if streaming else cyc.host_profile()  # (of the timed run)
    frame_source = None
    if streaming:
        # The same K frames handed over from host memory instead, each source from an IDENTICAL map state: a fresh map
        # and finder, the same warm-up frames, the same K timed frames (the run above continues on a map that has
        # grown, so its figure is not compared with these).  `value` stays the device-resident rate -- inputs in HBM
        # when the timed region starts, as the bench contract asks -- and says so in `value_inputs`.
        frame_source = {}
        t_py = timed_fleet_run(cyc.step, cyc.finish, args.steps, dist, torch.cuda.synchronize)
        host_loop = {"native_cpp_loop_cycles_per_s": fleet_value(n_gpus, args.steps, elapsed),
                     "python_ctypes_loop_cycles_per_s": fleet_value(n_gpus, args.steps, t_py)}
        for src in ("device", "pinned", "pageable"):
            c2 = GpuStreamCycle(map_size, box, frames, ctrl, device=local_rank, reference_order=args.reference_order)
            c2.frame_source = src
            c2.run_native(args.warmup + 8, args.serial_stages)
            c2.finish()
            t = timed_fleet_run(c2.prepare_native(args.steps, args.serial_stages), c2.finish, 1, dist,
                                torch.cuda.synchronize)
            frame_source[{"device": "device_resident", "pinned": "pinned_host", "pageable": "pageable_host"}[src]
                         + "_ms_per_frame"] = 1e3 * t / args.steps
            c2.close()
            del c2
        frame_source["note"] = ("three fresh maps, identical warm-up and timed frames per source; `value` is the "
                                "device-resident run of the main map (frames in HBM before the clock starts)")
    host_delivery = None
    if not streaming:
        cyc.map.profileEnable(0)
        t_py = timed_fleet_run(cyc.step, cyc.finish, args.steps, dist, torch.cuda.synchronize)
        host_loop = {"native_cpp_loop_cycles_per_s": fleet_value(n_gpus, args.steps, elapsed),
                     "python_ctypes_loop_cycles_per_s": fleet_value(n_gpus, args.steps, t_py)}
        # the same cycle with its results delivered to host containers every cycle (cells of every new cluster,
        # cost + gradient of every candidate): what a drop-in caller receives, beside the device-resident `value`
        if not args.serial_stages:
            dsec = [0.0, 0.0, 0.0]

            def _deliv():
                dsec[:] = run_delivered(cyc, args.steps)
            t_d = timed_fleet_run(_deliv, cyc.finish, 1, dist, torch.cuda.synchronize)
            host_delivery = {"cycles_per_s": fleet_value(n_gpus, args.steps, t_d),
                             "ms_per_step": 1e3 * t_d / args.steps,
                             "cells_d2h_ms": 1e3 * dsec[1] / args.steps, "cost_grad_d2h_ms": 1e3 * dsec[2] / args.steps,
                             "delivered": "cell lists of all new clusters (int32 addresses) + %d x %d cost/gradient doubles, "
                                          "every cycle, consumed one cycle behind the device: cycle k - 1's cells are copied "
                                          "from the retired buffer set (fuelmi_frontier_keep_previous) and its costs / gradients "
                                          "from the pinned slot the kernel wrote while cycle k runs"
                                          % (ctrl.shape[0], ctrl.shape[1] * 3 + 1)}
    dom_ms = dom_total_ms / max(n_launch, 1)  # mean over the timed region, as the contract asks

    fleet_cpu = None
    if n_gpus > 1 and not args.no_cpu_baseline:
        # fleet-vs-fleet: every rank times the CPU path on its own map, pinned to one core, concurrently
        cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else [0]
        core = cores[local_rank_raw % len(cores)]
        try:
            os.sched_setaffinity(0, {core})
        except Exception:
            pass
        dist.barrier()
        bud = min(args.cpu_budget, 8.0)
        if streaming:
            mine = cpu_baseline_stream_reference(map_size, box, frames, ctrl, bud) or cpu_baseline_stream(map_size, box, frames, ctrl, bud)
        else:
            mine = cpu_baseline_reference(map_size, box, occ, ctrl, bud) or cpu_baseline(map_size, box, occ, ctrl, bud)
        fleet_cpu = fleet_cpu_baseline(mine, rank, world, dist, core)
    if rank == 0:
        nv = cyc.map.nvox
        nvox = nv[0] * nv[1] * nv[2]
        if streaming:  # box-local stages: mean voxel count of the frames' local bounds
            nvox = float(np.mean(cyc.box_vox[-args.steps:])) if cyc.box_vox else 0.0
        # algorithmic HBM bytes per launch of each stage, box = nvox voxels
        fam_now = cyc.map.lastEsdfFamily() if args.esdf_family < 0 else args.esdf_family
        tmp_b = 2.0 if fam_now == 0 else 4.0  # y-pass result: 16-bit hand-over in the packed family (round 5), else u32
        alg_bytes = {
            "inflate": nvox * (3 / 8.0),          # occupied plane in, scratch plane out+in, inflated plane out
            "esdf_zy": nvox * (2 / 8.0 + tmp_b),  # inflated+unknown planes in, y-pass result out
            "esdf_x": nvox * (tmp_b + 4.0),       # y-pass result in, f32 distance out
            "bspline": ctrl.shape[0] * ctrl.shape[1] * 56.0,
        }
        achieved = alg_bytes[dominant] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters (collected in separate rocprofv3 --pmc passes,
        # corrected as MI355X_MICROARCH.md prescribes; committed under profiles/), else null
        traffic, traffic_commit = None, None
        try:
            pmc_doc = None
            for rnd in ("r06", "r05"):  # the newest committed counter pass of this workload
                pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic_%s.json" % (rnd, args.workload))
                if os.path.exists(pmc_path):
                    pmc_doc = json.load(open(pmc_path))
                    break
            pmc = pmc_doc["kernels"]
            traffic_commit = pmc_doc.get("commit")
            keys = {"esdf_zy": ("k_esdf_zy_pk2<", "k_esdf_zy_pk<", "k_esdf_zy4<"), "esdf_x": ("k_esdf_x_pk2<", "k_esdf_x_pk<", "k_esdf_x4"),
                    "inflate": ("k_inflate_fused", "k_inflate_yz"), "bspline": ("k_bspline_cost_grad",)}[dominant]
            hit = [v for key in keys for k, v in pmc.items() if key in k]
            traffic = hit[0]["hbm_bytes_per_launch"]
        except Exception:
            traffic, traffic_commit = None, None
        # The timed cycle overlaps the ESDF chain (map stream) with the frontier chain (own stream), so the
        # dominant kernel shares the CUs while it runs.  A short untimed pass with the two chains
        # serialised gives its duration in isolation (the number a kernel-level roofline usually quotes).
        iso_ms = iso_ms_all[dominant]
        out = {
            "metric": "plan_cycles_per_sec",
            "value": fleet_value(n_gpus, args.steps, elapsed),
            "unit": "cycles/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "timed_regions": ({"repetitions": len(regions), "reported": "median", "region_ms": [round(1e3 * r, 4) for r in regions]}
                              if not streaming else {"repetitions": 1, "reported": "the one region"}),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 bit-planes / u16 (u32 beyond 255 voxels) exact squared distances / f32 ESDF / f64 log-odds and B-spline",
            "data": "synthetic",
            "commit": os.environ.get("FUELMI_COMMIT"),  # git revision of the code (set by scripts/collect_profiles.sh)
            "config": {"workload": ("%s: %dx%dx%d @0.1m map per GPU, streaming 640x480 depth frames from an unknown "
                                    "map: device projection + fusion, box-local inflate+ESDF (mean box %.2f M voxels), "
                                    "incremental frontier search, %d B-spline candidates x 32 ctrl pts"
                                    % (args.workload, nv[0], nv[1], nv[2], nvox / 1e6, ctrl.shape[0])) if streaming else
                                   ("%s: %dx%dx%d @0.1m map per GPU, full-box inflate+ESDF, full "
                                    "exploration-box frontier search, %d B-spline candidates x 32 ctrl pts"
                                    % (args.workload, nv[0], nv[1], nv[2], ctrl.shape[0])),
                       "known_voxels": int(n_known), "frontier_clusters": int(cyc.n_clusters),
                       "reference_order": args.reference_order,
                       "esdf_family": {-1: "auto", 0: "plain (packed 16-bit z/y)", 1: "far-field", 2: "plain (32-bit z/y)"}[
                           cyc.map.lastEsdfFamily() if args.esdf_family < 0 else args.esdf_family],
                       "parallelism": "independent map per GPU (no collective)"},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_isolated": {k: round(v, 4) for k, v in iso_ms_all.items()},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": dom_ms, "algorithmic_bytes": alg_bytes[dominant],
                         # every single-kernel stage of the map chain, same accounting (isolated = chains serialised)
                         "kernels": {k: {"algorithmic_bytes": alg_bytes[k], "isolated_ms": round(iso_ms_all[k], 5),
                                         "isolated_frac": (alg_bytes[k] / (iso_ms_all[k] * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso_ms_all[k] > 0 else None}
                                     for k in ("inflate", "esdf_zy", "esdf_x", "bspline")},
                         # git revision the counter pass behind `traffic` was taken at (profiles/*.json carry it)
                         "traffic_commit": traffic_commit},
        }
        # whole-cycle figure: compulsory bytes of all stages (frontier: 4 planes of the search box in, 3 out,
        # plus ~40 B per frontier cell) x cycles/s against the HBM peak
        cyc_bytes = alg_bytes["inflate"] + alg_bytes["esdf_zy"] + alg_bytes["esdf_x"] + alg_bytes["bspline"] + \
            (nvox if streaming else nv[0] * nv[1] * nv[2]) * (7 / 8.0)
        cps_per_gpu = out["value"] / n_gpus
        # the stage that sets the cycle time: the longest chain of the two streams (frontier chain on its own
        # stream beside inflate -> ESDF -> B-spline on the map's), with its compulsory bytes
        fr_bytes = (nvox if streaming else nv[0] * nv[1] * nv[2]) * (7 / 8.0)
        map_chain_ms = stage_ms["inflate"] + stage_ms["esdf_zy"] + stage_ms["esdf_x"] + stage_ms["bspline"]
        map_chain_bytes = alg_bytes["inflate"] + alg_bytes["esdf_zy"] + alg_bytes["esdf_x"] + alg_bytes["bspline"]
        if stage_ms["frontier"] >= map_chain_ms:
            crit = {"stage": "frontier chain (predicate + tile CCL, cross-tile pairs, resolve, grouped output + flags)",
                    "kernels": 4, "algorithmic_bytes": fr_bytes, "ms": stage_ms["frontier"]}
        else:
            crit = {"stage": "map chain (inflate, ESDF z/y, ESDF x, B-spline batch)", "kernels": 4,
                    "algorithmic_bytes": map_chain_bytes, "ms": map_chain_ms}
        crit["achieved"] = crit["algorithmic_bytes"] / (crit["ms"] * 1e-3) / 1e9 if crit["ms"] > 0 else 0.0
        crit["frac"] = crit["achieved"] / HBM_PEAK_GBS
        crit["other_chain_ms"] = min(stage_ms["frontier"], map_chain_ms)
        out["roofline"]["critical"] = crit
        if timeline is not None:
            out["cycle_timeline"] = timeline
        if host_loop is not None:
            out["host_loop"] = host_loop
        if host_issue is not None:
            # what the ONE host thread that issues both streams spends per cycle inside each C-ABI call (us); everything
            # except search_end_polling is host work on the cycle's critical path
            host_issue["host_busy_us_per_cycle"] = round(sum(v for k, v in host_issue.items() if k != "search_end_polling")
                                                         - host_issue["search_end_polling"], 2)
            out["host_issue_us"] = host_issue
        if host_delivery is not None:
            out["host_delivery"] = host_delivery
        if streaming:
            out["value_inputs"] = "device-resident depth frames (in HBM before the timed region; see frame_source)"
        if frame_source is not None:
            out["frame_source"] = frame_source
        out["frontier_path"] = dict(zip(("fast", "legacy", "fallback"), cyc.ff.stats()))
        out["cycle_hbm"] = {"algorithmic_bytes_per_cycle": cyc_bytes, "achieved": cyc_bytes * cps_per_gpu / 1e9,
                            "unit": "GB/s", "frac": cyc_bytes * cps_per_gpu / 1e9 / HBM_PEAK_GBS}
        out["roofline"]["measured_triad_gbs"] = triad_gbs
        out["roofline"]["measured_expand_gbs"] = expand_gbs  # u16 in, f32 out: the packed x pass's traffic mix
        if iso_ms:
            out["roofline"]["isolated_launch_ms"] = iso_ms
            out["roofline"]["isolated_frac"] = alg_bytes[dominant] / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if fleet_cpu is not None:
            out["cpu_baseline"] = fleet_cpu
        if n_gpus == 1 and not args.no_cpu_baseline:
            if streaming:
                real = cpu_baseline_stream_reference(map_size, box, frames, ctrl, args.cpu_budget)
                port = cpu_baseline_stream(map_size, box, frames, ctrl, args.cpu_budget if real is None else
                                           min(args.cpu_budget, 6.0))
                if real is not None:
                    real["oracle_port"] = {k: port[k] for k in ("value", "sample")}
                out["cpu_baseline"] = real if real is not None else port
            else:
                # the reference's own code when its prebuilt library travelled with the repository, and the
                # oracle (the restatement, leaner: no std::function / per-call vectors) beside it
                real = cpu_baseline_reference(map_size, box, occ, ctrl, args.cpu_budget)
                port = cpu_baseline(map_size, box, occ, ctrl, args.cpu_budget if real is None else
                                    min(args.cpu_budget, 6.0))
                if real is not None:
                    real["oracle_port"] = {k: port[k] for k in ("value", "sample")}
                    out["cpu_baseline"] = real
                else:
                    out["cpu_baseline"] = port
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
