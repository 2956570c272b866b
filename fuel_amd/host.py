"""Python host mirror of the reference's C++ class interfaces over the libfuelmi C-ABI.

The production host layer is the C++ facade in fuel_amd/facade/ (same class names and
signatures as the reference).  These thin classes exist so tests and bench.py read like calls
into the reference: SDFMap (plan_env/include/plan_env/sdf_map.h:27-84), EDTEnvironment
(plan_env/include/plan_env/edt_environment.h:38-43), FrontierFinder
(active_perception/include/active_perception/frontier_finder.h:53-80) and BsplineOptimizer
(bspline_opt/include/bspline_opt/bspline_optimizer.h:36-59).  Everything computes on the GPU;
numpy arrays are only the host-side views the reference exposes (occupancy_buffer_ etc.).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BsplineBatch, BsplineCfg, FrontierCfg, MapCfg, MapInfo, check, lib)

# exploration.launch / algorithm.xml defaults (exploration_manager/launch/algorithm.xml:33-59,170-181)
DEFAULT_MAP = dict(resolution=0.1, ground_height=-1.0, obstacles_inflation=0.199,
                   local_bound_inflate=0.5, default_dist=0.0, optimistic=0, signed_dist=0,
                   p_hit=0.65, p_miss=0.35, p_min=0.12, p_max=0.90, p_occ=0.80,
                   max_ray_length=4.5, virtual_ceil_height=-10.0)
DEFAULT_BSPLINE = dict(ld_smooth=20.0, ld_dist=10.0, ld_feasi=2.0, ld_start=100.0, ld_end=0.5,
                       ld_guide=1.5, ld_waypt=0.3, ld_view=0.0, ld_time=1.0, dist0=0.7,
                       max_vel=2.0, max_acc=2.0, wnl=1.0, dlmin=0.0, bspline_degree=3)

SMOOTHNESS, DISTANCE, FEASIBILITY, START, END, GUIDE, WAYPOINTS, VIEWCONS, MINTIME = \
    (1 << k for k in range(9))
GUIDE_PHASE = SMOOTHNESS | GUIDE | START | END
NORMAL_PHASE = SMOOTHNESS | DISTANCE | FEASIBILITY | START | END


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


class SDFMap:
    """fast_planner::SDFMap with the grid resident in HBM."""
    UNKNOWN, FREE, OCCUPIED = 0, 1, 2
    ESDF_AUTO, ESDF_PLAIN, ESDF_FAR, ESDF_PLAIN32 = -1, 0, 1, 2
    # family every newly created map is pinned to (None: the library's per-update choice); the parity tests run each
    # ESDF test once per family by setting this
    default_esdf_family = None

    def __init__(self, map_size, box_min=None, box_max=None, device=0, **params):
        self.L = lib()
        p = dict(DEFAULT_MAP)
        p.update(params)
        c = MapCfg()
        for k, v in p.items():
            setattr(c, k, v)
        org = (-map_size[0] / 2.0, -map_size[1] / 2.0, p["ground_height"])
        bmin = box_min if box_min is not None else org
        bmax = box_max if box_max is not None else tuple(org[i] + map_size[i] for i in range(3))
        for i in range(3):
            c.map_size[i] = float(map_size[i])
            c.box_min[i] = float(bmin[i])
            c.box_max[i] = float(bmax[i])
        c.device = device
        self.cfg = c
        h = C.c_void_p()
        check(self.L.fuelmi_map_create(C.byref(c), C.byref(h)))
        self.h = h
        info = MapInfo()
        check(self.L.fuelmi_map_get_info(self.h, C.byref(info)))
        self.info = info
        self.nvox = tuple(info.voxel_num)
        self.N = self.nvox[0] * self.nvox[1] * self.nvox[2]
        self.origin = np.array(info.origin)
        self.res = c.resolution
        if SDFMap.default_esdf_family is not None:
            self.setEsdfFamily(SDFMap.default_esdf_family)

    def setEsdfFamily(self, family):
        """pin the ESDF kernel family (ESDF_AUTO / _PLAIN / _FAR / _PLAIN32); all are exact"""
        check(self.L.fuelmi_map_set_esdf_family(self.h, int(family)))

    def lastEsdfFamily(self):
        """family the z/y pass of the last updateESDF3d ran"""
        return self.L.fuelmi_map_last_esdf_family(self.h)

    def lastInflateKernel(self):
        """0: the fused inflation kernel ran in the last clearAndInflateLocalMap, 1: the factored pair"""
        return self.L.fuelmi_map_last_inflate_kernel(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.fuelmi_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- reference API ---
    def inputPointCloud(self, points, camera_pos):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        check(self.L.fuelmi_map_input_points(self.h, pts.ctypes.data, 12, len(pts), _d3(camera_pos)))

    @staticmethod
    def depthConfig(fx=387.229248046875, fy=387.229248046875, cx=321.04638671875, cy=243.44969177246094,
                    maxdist=5.0, mindist=0.2, margin=2, scaling=1000.0, skip=2):
        """map_ros/* parameters (exploration.launch:38-41, algorithm.xml:62-69)."""
        return _lib.DepthCfg(fx, fy, cx, cy, maxdist, mindist, margin, scaling, skip)

    def inputDepthImage(self, depth, camera_pos, camera_q_wxyz, cfg=None):
        """MapROS::depthPoseCallback's projection + fusion (map_ros.cpp:121-150,176-215) on the device.
        Returns proj_points_cnt."""
        img = np.ascontiguousarray(depth, dtype=np.uint16)
        cfg = cfg or self.depthConfig()
        n = C.c_int(0)
        q = (C.c_double * 4)(*[float(v) for v in camera_q_wxyz])
        check(self.L.fuelmi_map_input_depth(self.h, img.ctypes.data, img.shape[0], img.shape[1], C.byref(cfg),
                                            _d3(camera_pos), q, C.byref(n)))
        return n.value

    def inputDepthImageAt(self, ptr, rows, cols, camera_pos, camera_q_wxyz, cfg=None):
        """inputDepthImage for a frame addressed by a raw pointer: device memory (e.g. a torch CUDA tensor's
        data_ptr()), pinned / registered host memory, or pageable host memory.  The first two are read in place."""
        cfg = cfg or self.depthConfig()
        n = C.c_int(0)
        q = (C.c_double * 4)(*[float(v) for v in camera_q_wxyz])
        check(self.L.fuelmi_map_input_depth(self.h, C.c_void_p(int(ptr)), int(rows), int(cols), C.byref(cfg),
                                            _d3(camera_pos), q, C.byref(n)))
        return n.value

    def projectDepthImage(self, depth, camera_pos, camera_q_wxyz, cfg=None):
        """MapROS::proessDepthImage only: the projected world points (float32 [n,3])."""
        img = np.ascontiguousarray(depth, dtype=np.uint16)
        cfg = cfg or self.depthConfig()
        cap = img.shape[0] * img.shape[1]
        out = np.empty((cap, 3), dtype=np.float32)
        n = C.c_int(0)
        q = (C.c_double * 4)(*[float(v) for v in camera_q_wxyz])
        check(self.L.fuelmi_map_project_depth(self.h, img.ctypes.data, img.shape[0], img.shape[1], C.byref(cfg),
                                              _d3(camera_pos), q, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value].copy()

    def clearAndInflateLocalMap(self):
        check(self.L.fuelmi_map_inflate_local(self.h))

    def updateESDF3d(self):
        check(self.L.fuelmi_map_update_esdf(self.h))

    def resetBuffer(self, lo=None, hi=None):
        if lo is None:
            check(self.L.fuelmi_map_reset_buffer_all(self.h))
        else:
            check(self.L.fuelmi_map_reset_buffer(self.h, _d3(lo), _d3(hi)))

    def setOccupied(self, pos, occ=1):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        check(self.L.fuelmi_map_set_occupied(self.h, _dp(pos), len(pos), occ))

    def getDistWithGrad(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        d = np.empty(len(pos))
        g = np.empty((len(pos), 3))
        check(self.L.fuelmi_map_dist_grad(self.h, _dp(pos), len(pos), _dp(d), _dp(g)))
        return d, g

    def getDistance(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        d = np.empty(len(pos))
        check(self.L.fuelmi_map_coarse_dist(self.h, _dp(pos), len(pos), _dp(d)))
        return d

    def getOccupancy(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, 3)
        o = np.empty(len(idx), dtype=np.int32)
        i = np.empty(len(idx), dtype=np.int32)
        check(self.L.fuelmi_map_query_state(self.h, _ip(idx), len(idx), _ip(o), _ip(i)))
        return o, i

    def getUpdatedBox(self, reset=False):
        a, b = (C.c_double * 3)(), (C.c_double * 3)()
        check(self.L.fuelmi_map_get_updated_box(self.h, a, b, int(reset)))
        return np.array(a), np.array(b)

    def setUpdatedBox(self, lo, hi):
        check(self.L.fuelmi_map_set_updated_box(self.h, _d3(lo), _d3(hi)))

    def getLocalBound(self):
        a, b = (C.c_int * 3)(), (C.c_int * 3)()
        check(self.L.fuelmi_map_get_local_bound(self.h, a, b))
        return tuple(a), tuple(b)

    def setLocalBound(self, lo, hi):
        check(self.L.fuelmi_map_set_local_bound(self.h, _i3(lo), _i3(hi)))

    def getBoxIndex(self):
        return tuple(self.info.box_min), tuple(self.info.box_max)

    # --- device <-> host ---
    def uploadOccupancy(self, occ):
        occ = np.ascontiguousarray(occ, dtype=np.float64).reshape(-1)
        assert occ.size == self.N
        check(self.L.fuelmi_map_upload_occupancy(self.h, _dp(occ)))

    def syncHost(self, occupancy=False, inflate=False, distance=False, box=None):
        """Refresh host mirrors (occupancy_buffer_, occupancy_buffer_inflate_, distance_buffer_)."""
        out = {}
        o = np.zeros(self.N) if occupancy else None
        i = np.zeros(self.N, dtype=np.int8) if inflate else None
        d = np.zeros(self.N) if distance else None
        bmin = _i3(box[0]) if box else None
        bmax = _i3(box[1]) if box else None
        check(self.L.fuelmi_map_sync_host(self.h, bmin, bmax, _dp(o),
                                          None if i is None else i.ctypes.data, _dp(d)))
        if occupancy:
            out["occupancy"] = o
        if inflate:
            out["inflate"] = i
        if distance:
            out["distance"] = d
        return out

    def synchronize(self):
        check(self.L.fuelmi_map_synchronize(self.h))

    # --- measurement ---
    def timerBegin(self):
        check(self.L.fuelmi_timer_begin(self.h))

    def timerEnd(self):
        ms = C.c_float()
        check(self.L.fuelmi_timer_end(self.h, C.byref(ms)))
        return ms.value

    def profileEnable(self, mask):
        check(self.L.fuelmi_profile_enable(self.h, mask))

    def profileGet(self, stage):
        n = C.c_int()
        t = C.c_double()
        check(self.L.fuelmi_profile_get(self.h, stage, C.byref(n), C.byref(t)))
        return n.value, t.value

    def profileTimeline(self, stage, cap=4096):
        """(begin, end) of every bracket of a stage in ms after the profileEnable call that armed it"""
        a = np.empty(cap, dtype=np.float64)
        b = np.empty(cap, dtype=np.float64)
        n = C.c_int()
        check(self.L.fuelmi_profile_get_timeline(self.h, stage, _dp(a), _dp(b), cap, C.byref(n)))
        return a[:n.value].copy(), b[:n.value].copy()

    def profileSamples(self, stage, cap=4096):
        """Per-launch milliseconds of a stage since profileEnable."""
        ms = np.empty(cap)
        n = C.c_int()
        check(self.L.fuelmi_profile_get_samples(self.h, stage, _dp(ms), cap, C.byref(n)))
        return ms[:n.value].copy()


class DeviceBuffer:
    """A numpy array's bytes in device memory (fuelmi_device_alloc / _upload): .ptr is the device address."""

    def __init__(self, array, device=0):
        a = np.ascontiguousarray(array)
        self.L = lib()
        p = C.c_void_p()
        check(self.L.fuelmi_device_alloc(int(device), a.nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, a.nbytes
        check(self.L.fuelmi_device_upload(C.c_void_p(self.ptr), a.ctypes.data, a.nbytes))

    def close(self):
        if getattr(self, "ptr", None):
            self.L.fuelmi_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RegisteredHostBuffer:
    """A numpy array registered with the device (fuelmi_host_register: a camera driver's frame ring, read in place by
    the fusion kernels).  The registration is undone BEFORE the array's memory goes back to the allocator -- by close()
    or, at the latest, when this object dies: a registration that outlives its memory stays in the HIP runtime's
    address map, and a later pageable copy from whatever the allocator puts there fails with "invalid argument"."""

    def __init__(self, array):
        self.array = np.ascontiguousarray(array)
        self.L = lib()
        check(self.L.fuelmi_host_register(self.array.ctypes.data, self.array.nbytes))
        self.ptr, self.nbytes = self.array.ctypes.data, self.array.nbytes

    def close(self):
        if getattr(self, "ptr", None):
            self.L.fuelmi_host_unregister(C.c_void_p(self.ptr))
            self.ptr = None
        self.array = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EDTEnvironment:
    """fast_planner::EDTEnvironment: distance/gradient query facade over SDFMap."""

    def __init__(self):
        self.sdf_map_ = None

    def setMap(self, sdf_map):
        self.sdf_map_ = sdf_map

    def evaluateEDTWithGrad(self, pos, time=-1.0):
        return self.sdf_map_.getDistWithGrad(pos)

    def evaluateCoarseEDT(self, pos, time=-1.0):
        return self.sdf_map_.getDistance(pos)


class FrontierFinder:
    """Grid part of fast_planner::FrontierFinder (searchFrontiers / expandFrontier)."""

    def __init__(self, edt_or_map, cluster_min=100, min_z=0.4, cluster_size_xy=2.0, down_sample=3, split=False,
                 reference_order=False):
        """split=True: searchFrontiers ends with splitLargeFrontiers (frontier_finder.cpp:120,166-242) and
        every new cluster carries its down-sampled filtered_cells_.  reference_order=True: cells in the
        reference's BFS order, means / VoxelGrid centroids summed in that order (bit-exact against the reference;
        default: ascending voxel address, order-free means)."""
        self.L = lib()
        self.map = edt_or_map.sdf_map_ if isinstance(edt_or_map, EDTEnvironment) else edt_or_map
        cfg = FrontierCfg(cluster_min, min_z, cluster_size_xy, down_sample, int(split), int(reference_order))  # (True -> 1; pass 2 for "auto")
        h = C.c_void_p()
        check(self.L.fuelmi_frontier_create(self.map.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.fuelmi_frontier_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def searchFrontiers(self):
        n = C.c_int()
        check(self.L.fuelmi_frontier_search(self.h, C.byref(n)))
        return n.value

    def searchFrontiersBegin(self):
        check(self.L.fuelmi_frontier_search_begin(self.h))

    def searchFrontiersEnd(self):
        n = C.c_int()
        check(self.L.fuelmi_frontier_search_end(self.h, C.byref(n)))
        return n.value

    def commit(self, dormant=False):
        check(self.L.fuelmi_frontier_commit(self.h, int(dormant)))

    def reset(self):
        check(self.L.fuelmi_frontier_reset(self.h))

    def sync(self):
        """wait for everything the finder has queued (incl. the regrouping tail of the last search)"""
        from ._lib import check as _c
        _c(self.L.fuelmi_frontier_synchronize(self.h))

    def stats(self):
        """(searches on the fast chain, on the legacy chain, fast-chain searches that fell back)"""
        o = (C.c_int * 3)()
        check(self.L.fuelmi_frontier_stats(self.h, o))
        return tuple(o)

    def resolvedInLaunch(self):
        """fast-chain searches resolved by the last workgroup of k_tile_cross itself (the others: by k_resolve)"""
        return self.L.fuelmi_frontier_resolved_in_launch(self.h)

    def orderStats(self):
        """(order of the last search: 0 address / 1 reference BFS, searches in the reference's order, mode-2 searches
        that fell back to the address order, cells of the cluster that forced the last fallback)"""
        o = (C.c_int * 4)()
        check(self.L.fuelmi_frontier_order_stats(self.h, o))
        return tuple(o)

    def clusters(self, which=0):
        out = []
        cnt = self.L.fuelmi_frontier_count(self.h, which)
        if cnt < 0:
            check(cnt)
        for k in range(cnt):
            n = self.L.fuelmi_frontier_cluster_size(self.h, which, k)
            a = np.empty(n, dtype=np.int32)
            check(self.L.fuelmi_frontier_cluster_cells(self.h, which, k, _ip(a)))
            out.append(a)
        return out

    def clusterCentres(self, which=0):
        """cells_ of every cluster as voxel centres ([n, 3] doubles each), decoded by the library"""
        out = []
        cnt = self.L.fuelmi_frontier_count(self.h, which)
        if cnt < 0:
            check(cnt)
        for k in range(cnt):
            n = self.L.fuelmi_frontier_cluster_size(self.h, which, k)
            a = np.empty((n, 3))
            check(self.L.fuelmi_frontier_cluster_centres(self.h, which, k, _dp(a)))
            out.append(a)
        return out

    def keepPrevious(self, on=True):
        """fuelmi_frontier_keep_previous: a reset keeps the retired search's new clusters readable as list 3"""
        check(self.L.fuelmi_frontier_keep_previous(self.h, int(bool(on))))

    @staticmethod
    def viewpointConfig(rmin=1.5, rmax=2.5, rnum=3, dphi=15 * 3.1415926 / 180.0, clearance=0.21, min_visib_num=15,
                        min_candidate_dist=0.75, min_view_finish_fraction=0.2, top_angle=0.56125,
                        left_angle=0.69222, right_angle=0.68901, max_dist=4.5):
        """frontier/candidate_* and perception_utils/* of algorithm.xml:106-121."""
        return _lib.ViewpointCfg(rmin, rmax, rnum, dphi, clearance, min_visib_num, min_candidate_dist,
                                 min_view_finish_fraction, top_angle, left_angle, right_angle, max_dist)

    def setViewpointConfig(self, cfg):
        self.vcfg = cfg
        check(self.L.fuelmi_frontier_set_viewpoint_cfg(self.h, C.byref(cfg)))

    def computeFrontiersToVisit(self):
        """sampleViewpoints for the new clusters, then frontiers_ / dormant_frontiers_ (reference :392-423).
        Returns (new active, new dormant)."""
        a, d = C.c_int(), C.c_int()
        check(self.L.fuelmi_frontier_compute_to_visit(self.h, C.byref(a), C.byref(d)))
        return a.value, d.value

    def viewpoints(self, which, k):
        """(pos_yaw [n,4], visib_num [n]) of cluster k, best coverage first."""
        n = self.L.fuelmi_frontier_viewpoint_count(self.h, which, k)
        if n < 0:
            check(n)
        py = np.empty((n, 4))
        vis = np.empty(n, dtype=np.int32)
        if n:
            check(self.L.fuelmi_frontier_viewpoints(self.h, which, k, _dp(py), _ip(vis)))
        return py, vis

    def isFrontierCovered(self):
        c = C.c_int()
        check(self.L.fuelmi_frontier_is_covered(self.h, C.byref(c)))
        return bool(c.value)

    def getTopViewpointsInfo(self, cur_pos):
        """reference :425-450: per active frontier the best viewpoint farther than min_candidate_dist."""
        cur = np.asarray(cur_pos, dtype=float)
        pts, yaws, avgs = [], [], []
        for k in range(self.L.fuelmi_frontier_count(self.h, 1)):
            py, _ = self.viewpoints(1, k)
            pick = py[0]
            for v in py:
                if np.linalg.norm(v[:3] - cur) < self.vcfg.min_candidate_dist:
                    continue
                pick = v
                break
            pts.append(pick[:3].copy())
            yaws.append(float(pick[3]))
            avgs.append(self.clusterInfo(1, k)[0])
        return pts, yaws, avgs

    def filtered(self, which, k):
        """Frontier::filtered_cells_ of cluster k: float32 [n,3] (split=True searches only)."""
        n = self.L.fuelmi_frontier_cluster_filtered_size(self.h, which, k)
        if n < 0:
            check(n)
        out = np.empty((n, 3), dtype=np.float32)
        if n:
            check(self.L.fuelmi_frontier_cluster_filtered(self.h, which, k, out.ctypes.data))
        return out

    def getFrontiers(self):
        """clusters of frontiers_ as voxel-centre positions (reference: getFrontiers)."""
        nv = self.map.nvox
        res = []
        for a in self.clusters(1):
            idx = np.stack(np.unravel_index(a, nv), axis=1)
            res.append((idx + 0.5) * self.map.res + self.map.origin)
        return res

    def clusterInfo(self, which, k):
        o = np.empty(9)
        check(self.L.fuelmi_frontier_cluster_info(self.h, which, k, _dp(o)))
        return o[:3], o[3:6], o[6:9]

    def removedIds(self):
        n = self.L.fuelmi_frontier_removed_count(self.h)
        a = np.empty(max(n, 0), dtype=np.int32)
        if n > 0:
            check(self.L.fuelmi_frontier_removed_ids(self.h, _ip(a)))
        return a

    def flags(self):
        f = np.zeros(self.map.N, dtype=np.int8)
        check(self.L.fuelmi_frontier_get_flags(self.h, f.ctypes.data))
        return f


class BsplineBatchProblem:
    """Keeps the numpy arrays of one batch alive and exposes the C struct."""

    def __init__(self, x, point_num, cost_function, pt_dist, start_state=None, end_state=None, end_n=3,
                 dim=3, knot_span=None, time_lb=None, guide_pts=None, waypoints=None, waypt_idx=None,
                 view_pt=None, view_dir=None, view_idx=None):
        f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        self.x = f64(x)
        self.C = self.x.shape[0]
        self.nvar = self.x.shape[1]
        self.pt_dist = f64(np.broadcast_to(pt_dist, (self.C,)))
        self.knot_span = f64(np.broadcast_to(knot_span if knot_span is not None else 0.0, (self.C,)))
        self.time_lb = None if time_lb is None else f64(np.broadcast_to(time_lb, (self.C,)))
        self.start_state = f64(start_state)
        self.end_state = f64(end_state)
        self.guide_pts = f64(guide_pts)
        self.waypoints = f64(waypoints)
        self.waypt_idx = i32(waypt_idx)
        self.view_pt = f64(view_pt)
        self.view_dir = f64(view_dir)
        self.view_idx = i32(view_idx)
        b = BsplineBatch()
        b.cost_function = cost_function
        b.dim = dim
        b.point_num = point_num
        b.n_traj = self.C
        b.x = _dp(self.x)
        b.pt_dist = _dp(self.pt_dist)
        b.knot_span = _dp(self.knot_span)
        b.time_lb = _dp(self.time_lb)
        b.start_state = _dp(self.start_state)
        b.end_state = _dp(self.end_state)
        b.end_n = end_n
        b.guide_pts = _dp(self.guide_pts)
        b.waypoints = _dp(self.waypoints)
        b.waypt_idx = _ip(self.waypt_idx)
        b.n_waypt = 0 if self.waypoints is None else self.waypoints.shape[1]
        b.view_pt = _dp(self.view_pt)
        b.view_dir = _dp(self.view_dir)
        b.view_idx = _ip(self.view_idx)
        self.c = b


class BsplineOptimizer:
    """Cost/gradient side of fast_planner::BsplineOptimizer, batched over candidates."""

    def __init__(self, **params):
        p = dict(DEFAULT_BSPLINE)
        p.update(params)
        self.cfg = BsplineCfg(**p)
        self.L = lib()
        self.env = None

    def setEnvironment(self, env):
        self.env = env

    def _map(self):
        return self.env.sdf_map_ if isinstance(self.env, EDTEnvironment) else self.env

    def combineCost(self, problem):
        """Evaluate C trajectories; returns (cost[C], grad[C, nvar])."""
        cost = np.empty(problem.C)
        grad = np.empty((problem.C, problem.nvar))
        check(self.L.fuelmi_bspline_cost_grad(self._map().h, C.byref(self.cfg), C.byref(problem.c),
                                              _dp(cost), _dp(grad)))
        return cost, grad

    def optimize(self, problem, max_eval=300, max_time=-1.0):
        """BsplineOptimizer::optimize() for the C trajectories of `problem` in ONE call (fuelmi_bspline_optimize: a
        query slot of the map -- no device allocation, re-entrant).  Returns (x [C, nvar], cost [C], evals [C])."""
        x = np.empty((problem.C, problem.nvar))
        cost = np.empty(problem.C)
        ev = np.empty(problem.C, dtype=np.int32)
        check(self.L.fuelmi_bspline_optimize(self._map().h, C.byref(self.cfg), C.byref(problem.c), int(max_eval),
                                             float(max_time), _dp(x), _dp(cost), _ip(ev)))
        return x, cost, ev

    def deviceProblem(self, problem):
        return BsplineDeviceProblem(self, problem)


class NonUniformBspline:
    """The two NonUniformBspline calls the planners wrap around optimize() (bspline/src/non_uniform_bspline.cpp),
    batched over candidates on the map's device."""

    @staticmethod
    def parameterizeToBspline(sdf_map, ts, points, derivs, degree=3):
        """ts [C], points [C][K][3], derivs [C][4][3] -> control points [C][K+degree-1][3] (:178-265)."""
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        points = np.ascontiguousarray(points, dtype=np.float64)
        derivs = np.ascontiguousarray(derivs, dtype=np.float64)
        if points.ndim != 3 or points.shape[2] != 3 or ts.shape != (points.shape[0],) or \
                derivs.shape != (points.shape[0], 4, 3):
            raise ValueError("parameterizeToBspline: ts [C], points [C][K][3], derivs [C][4][3]")
        cn, k = points.shape[0], points.shape[1]
        ctrl = np.empty((cn, k + degree - 1, 3))
        check(lib().fuelmi_bspline_parameterize(sdf_map.h, cn, k, int(degree), _dp(ts), _dp(points), _dp(derivs),
                                                _dp(ctrl)))
        return ctrl

    @staticmethod
    def getBoundaryStates(sdf_map, ctrl, ts, degree=3, ks=2, ke=0):
        """ctrl [C][N][3], ts [C] -> (start [C][ks+1][3], end [C][ke+1][3]) (:107-122)."""
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        if ctrl.ndim != 3 or ctrl.shape[2] != 3 or ts.shape != (ctrl.shape[0],):
            raise ValueError("getBoundaryStates: ctrl [C][N][3], ts [C]")
        cn = ctrl.shape[0]
        start = np.empty((cn, ks + 1, 3))
        end = np.empty((cn, ke + 1, 3))
        check(lib().fuelmi_bspline_boundary_states(sdf_map.h, cn, ctrl.shape[1], int(degree), _dp(ts), _dp(ctrl),
                                                   int(ks), int(ke), _dp(start), _dp(end)))
        return start, end


class BsplineDeviceProblem:
    def __init__(self, opt, problem):
        self.L = opt.L
        self.problem = problem
        self.map = opt._map()
        h = C.c_void_p()
        check(self.L.fuelmi_bspline_dev_create(self.map.h, C.byref(opt.cfg), C.byref(problem.c), C.byref(h)))
        self.h = h

    def eval(self):
        check(self.L.fuelmi_bspline_dev_eval(self.h))

    def evalPinned(self, slot):
        check(self.L.fuelmi_bspline_dev_eval_pinned(self.h, int(slot)))

    def collect(self, slot):
        cost = np.empty(self.problem.C)
        grad = np.empty((self.problem.C, self.problem.nvar))
        check(self.L.fuelmi_bspline_dev_collect(self.h, int(slot), _dp(cost), _dp(grad)))
        return cost, grad

    def download(self):
        cost = np.empty(self.problem.C)
        grad = np.empty((self.problem.C, self.problem.nvar))
        check(self.L.fuelmi_bspline_dev_download(self.h, _dp(cost), _dp(grad)))
        return cost, grad

    def optimize(self, max_eval=300, max_time=-1.0):
        """BsplineOptimizer::optimize() for every candidate, on the device; max_time (seconds, <= 0: none) is the
        solver's wall-clock cap (set_maxtime).  Returns (best_x [C][nvar], best_cost [C], evaluations [C])."""
        c = self.problem.c
        nvar = c.dim * c.point_num + (1 if c.cost_function & MINTIME else 0)
        x = np.empty((c.n_traj, nvar))
        cost = np.empty(c.n_traj)
        ev = np.empty(c.n_traj, dtype=np.int32)
        check(self.L.fuelmi_bspline_dev_optimize_timed(self.h, int(max_eval), float(max_time), _dp(x), _dp(cost), _ip(ev)))
        return x, cost, ev

    def loadSamples(self, ts, points, derivs):
        """samples -> parameterizeToBspline -> getBoundaryStates(2, 0) -> setBoundaryStates + pt_dist_ on the
        device (planner_manager.cpp:161-184): ts [C], points [C][K][3], derivs [C][4][3]."""
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        points = np.ascontiguousarray(points, dtype=np.float64)
        derivs = np.ascontiguousarray(derivs, dtype=np.float64)
        c = self.problem.c
        if ts.shape != (c.n_traj,) or points.ndim != 3 or points.shape[0] != c.n_traj or points.shape[2] != 3 \
                or derivs.shape != (c.n_traj, 4, 3):
            raise ValueError("loadSamples: ts [C], points [C][K][3], derivs [C][4][3]")
        check(self.L.fuelmi_bspline_dev_load_samples(self.h, points.shape[1], _dp(ts), _dp(points), _dp(derivs)))

    def close(self):
        if getattr(self, "h", None):
            self.L.fuelmi_bspline_dev_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
