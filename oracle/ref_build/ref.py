"""ctypes binding of oracle/_ref/libfuel_ref.so: the REAL reference classes (SDFMap,
EDTEnvironment, RayCaster, BsplineOptimizer) compiled from /root/reference with header stand-ins.
TEST INFRASTRUCTURE ONLY.  available() is False where the library was never built."""
import ctypes as C
import os

import numpy as np

from oracle import fuel_oracle as fo

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.path.dirname(_HERE), "_ref", "libfuel_ref.so")
_LIB = None


def available():
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(SO)
        P = C.c_void_p
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.ref_map_create.restype = P
        L.ref_map_create.argtypes = [C.POINTER(fo.MapCfg)]
        L.ref_map_destroy.argtypes = [P]
        L.ref_map_voxel_num.argtypes = [P, ip]
        for n in ("ref_map_occupancy", "ref_map_distance"):
            getattr(L, n).restype = dp
            getattr(L, n).argtypes = [P]
        L.ref_map_inflate_buf.restype = C.POINTER(C.c_char)
        L.ref_map_inflate_buf.argtypes = [P]
        L.ref_map_input_points.argtypes = [P, C.c_void_p, C.c_int, dp]
        L.ref_map_inflate_local.argtypes = [P]
        L.ref_map_update_esdf.argtypes = [P]
        L.ref_map_get_local_bound.argtypes = [P, ip, ip]
        L.ref_map_set_local_bound.argtypes = [P, ip, ip]
        L.ref_map_get_updated_box.argtypes = [P, dp, dp, C.c_int]
        L.ref_map_reset_buffer_all.argtypes = [P]
        L.ref_map_set_occupied.argtypes = [P, dp, C.c_int]
        L.ref_map_get_occupancy_idx.argtypes = [P, ip]
        L.ref_map_dist_grad.argtypes = [P, dp, C.c_int, dp, dp]
        L.ref_raycast_cells.argtypes = [P, dp, dp, ip, C.c_int]
        L.ref_bspline_cost_grad.argtypes = [P, C.POINTER(fo.BsplineCfg), C.POINTER(fo.BsplineProblem), dp, dp, dp]
        L.ref_map_set_updated_box.argtypes = [P, dp, dp]
        L.ref_frontier_create.restype = P
        L.ref_frontier_create.argtypes = [P, C.c_int, C.c_double]
        L.ref_frontier_destroy.argtypes = [P]
        L.ref_frontier_flags.restype = C.POINTER(C.c_char)
        L.ref_frontier_flags.argtypes = [P]
        L.ref_frontier_search.argtypes = [P]
        L.ref_frontier_commit.argtypes = [P, C.c_int]
        L.ref_frontier_count.argtypes = [P, C.c_int]
        L.ref_frontier_cluster_size.argtypes = [P, C.c_int, C.c_int]
        L.ref_frontier_cluster_cells.argtypes = [P, C.c_int, C.c_int, ip]
        L.ref_frontier_cluster_info.argtypes = [P, C.c_int, C.c_int, dp]
        L.ref_frontier_create_full.restype = P
        L.ref_frontier_create_full.argtypes = [P, C.c_int, C.c_double, dp]
        L.ref_frontier_compute_to_visit.argtypes = [P]
        L.ref_frontier_is_covered.argtypes = [P]
        L.ref_frontier_viewpoint_count.argtypes = [P, C.c_int, C.c_int]
        L.ref_frontier_viewpoints.argtypes = [P, C.c_int, C.c_int, dp, ip]
        L.ref_frontier_cluster_filtered_size.argtypes = [P, C.c_int, C.c_int]
        L.ref_frontier_cluster_filtered.argtypes = [P, C.c_int, C.c_int, dp]
        L.ref_frontier_removed_count.argtypes = [P]
        L.ref_frontier_removed_ids.argtypes = [P, ip]
        _LIB = L
    return _LIB


class RefFrontier:
    """fast_planner::FrontierFinder (the reference's own searchFrontiers / expandFrontier)."""

    def __init__(self, rmap, cluster_min=100, cluster_size_xy=-1.0, viewpoint_cfg=None):
        """cluster_size_xy < 0: splitLargeFrontiers never splits (region-grown clusters only).
        viewpoint_cfg: fo.ViewpointCfg -> the frontier/candidate_* and perception_utils/* parameters."""
        self.L = lib()
        self.map = rmap
        if viewpoint_cfg is None:
            self.h = self.L.ref_frontier_create(rmap.h, cluster_min, C.c_double(cluster_size_xy))
        else:
            v = viewpoint_cfg
            vp = np.array([v.candidate_rmin, v.candidate_rmax, v.candidate_rnum, v.candidate_dphi,
                           v.min_candidate_clearance, v.min_visib_num, v.min_candidate_dist,
                           v.min_view_finish_fraction, v.top_angle, v.left_angle, v.right_angle, v.max_dist],
                          dtype=np.float64)
            self.h = self.L.ref_frontier_create_full(rmap.h, cluster_min, C.c_double(cluster_size_xy), fo._dp(vp))
        self.flags = np.ctypeslib.as_array(C.cast(self.L.ref_frontier_flags(self.h), C.POINTER(C.c_int8)),
                                           shape=(rmap.N,))

    def __del__(self):
        try:
            self.L.ref_frontier_destroy(self.h)
        except Exception:
            pass

    def search(self):
        return self.L.ref_frontier_search(self.h)

    def commit(self, dormant=False):
        self.L.ref_frontier_commit(self.h, int(dormant))

    def clusters(self, which=0):
        out = []
        for k in range(self.L.ref_frontier_count(self.h, which)):
            n = self.L.ref_frontier_cluster_size(self.h, which, k)
            a = np.empty(n, dtype=np.int32)
            self.L.ref_frontier_cluster_cells(self.h, which, k, fo._ip(a))
            out.append(a)
        return out

    def cluster_info(self, which, k):
        o = np.empty(9)
        self.L.ref_frontier_cluster_info(self.h, which, k, fo._dp(o))
        return o[:3], o[3:6], o[6:9]

    def compute_to_visit(self):
        self.L.ref_frontier_compute_to_visit(self.h)

    def is_covered(self):
        return bool(self.L.ref_frontier_is_covered(self.h))

    def viewpoints(self, which, k):
        n = self.L.ref_frontier_viewpoint_count(self.h, which, k)
        py = np.empty((n, 4))
        vis = np.empty(n, dtype=np.int32)
        if n:
            self.L.ref_frontier_viewpoints(self.h, which, k, fo._dp(py), fo._ip(vis))
        return py, vis

    def filtered(self, which, k):
        n = self.L.ref_frontier_cluster_filtered_size(self.h, which, k)
        o = np.empty((n, 3))
        if n:
            self.L.ref_frontier_cluster_filtered(self.h, which, k, fo._dp(o))
        return o

    def update_cost_matrix(self):
        self.L.ref_frontier_update_cost_matrix.restype = None
        self.L.ref_frontier_update_cost_matrix.argtypes = [C.c_void_p]
        self.L.ref_frontier_update_cost_matrix(self.h)

    def full_cost_matrix(self, pos, vel=(0.0, 0.0, 0.0), yaw=(0.0, 0.0, 0.0)):
        D = C.POINTER(C.c_double)
        self.L.ref_frontier_full_cost_matrix.restype = C.c_int
        self.L.ref_frontier_full_cost_matrix.argtypes = [C.c_void_p, D, D, D, D, C.c_int]
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (pos, vel, yaw)]
        buf = np.zeros(1 << 16)
        d = self.L.ref_frontier_full_cost_matrix(self.h, *[v.ctypes.data_as(D) for v in a], buf.ctypes.data_as(D), buf.size)
        assert d > 0
        return buf[:d * d].reshape(d, d).copy()

    def path_for_tour(self, pos, ids):
        D = C.POINTER(C.c_double)
        self.L.ref_frontier_path_for_tour.restype = C.c_int
        self.L.ref_frontier_path_for_tour.argtypes = [C.c_void_p, D, C.POINTER(C.c_int), C.c_int, D, C.c_int]
        p = np.ascontiguousarray(pos, dtype=np.float64)
        i = np.ascontiguousarray(ids, dtype=np.int32)
        buf = np.zeros(3 * 4096)
        n = self.L.ref_frontier_path_for_tour(self.h, p.ctypes.data_as(D), i.ctypes.data_as(C.POINTER(C.c_int)), len(i),
                                              buf.ctypes.data_as(D), 4096)
        assert n >= 0
        return buf[:3 * n].reshape(n, 3).copy()

    def removed_ids(self):
        n = self.L.ref_frontier_removed_count(self.h)
        a = np.empty(n, dtype=np.int32)
        if n:
            self.L.ref_frontier_removed_ids(self.h, fo._ip(a))
        return a


class RefMap:
    """Same surface as fuel_oracle.OracleMap, backed by the reference's own SDFMap."""

    def __init__(self, map_size, box_min=None, box_max=None, **kw):
        self.L = lib()
        self.cfg = fo.make_cfg(fo.MapCfg, map_size, box_min, box_max, **kw)
        self.h = self.L.ref_map_create(C.byref(self.cfg))
        nv = (C.c_int * 3)()
        self.L.ref_map_voxel_num(self.h, nv)
        self.nvox = tuple(nv)
        self.N = self.nvox[0] * self.nvox[1] * self.nvox[2]
        self.occ = np.ctypeslib.as_array(self.L.ref_map_occupancy(self.h), shape=(self.N,))
        self.dist = np.ctypeslib.as_array(self.L.ref_map_distance(self.h), shape=(self.N,))
        self.infl = np.ctypeslib.as_array(C.cast(self.L.ref_map_inflate_buf(self.h), C.POINTER(C.c_int8)),
                                          shape=(self.N,))

    def __del__(self):
        try:
            self.L.ref_map_destroy(self.h)
        except Exception:
            pass

    def input_points(self, pts, cam):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        self.L.ref_map_input_points(self.h, pts.ctypes.data, len(pts), fo._d3(cam))

    def inflate_local(self):
        self.L.ref_map_inflate_local(self.h)

    def update_esdf(self):
        self.L.ref_map_update_esdf(self.h)

    def get_local_bound(self):
        a, b = (C.c_int * 3)(), (C.c_int * 3)()
        self.L.ref_map_get_local_bound(self.h, a, b)
        return tuple(a), tuple(b)

    def set_local_bound(self, lo, hi):
        self.L.ref_map_set_local_bound(self.h, fo._i3(lo), fo._i3(hi))

    def get_updated_box(self, reset=False):
        a, b = (C.c_double * 3)(), (C.c_double * 3)()
        self.L.ref_map_get_updated_box(self.h, a, b, int(reset))
        return np.array(a), np.array(b)

    def set_updated_box(self, lo, hi):
        self.L.ref_map_set_updated_box(self.h, fo._d3(lo), fo._d3(hi))

    def reset_buffer(self):
        self.L.ref_map_reset_buffer_all(self.h)

    def set_occupied(self, pos, occ=1):
        self.L.ref_map_set_occupied(self.h, fo._d3(pos), occ)

    def get_occupancy_idx(self, idx):
        return self.L.ref_map_get_occupancy_idx(self.h, fo._i3(idx))

    def dist_grad(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        d = np.empty(len(pos))
        g = np.empty((len(pos), 3))
        self.L.ref_map_dist_grad(self.h, fo._dp(pos), len(pos), fo._dp(d), fo._dp(g))
        return d, g

    def raycast_cells(self, start, end, cap=4096):
        out = np.empty((cap, 3), dtype=np.int32)
        n = self.L.ref_raycast_cells(self.h, fo._d3(start), fo._d3(end), fo._ip(out), cap)
        return out[:min(n, cap)].copy()


def bspline_cost_grad(rmap, x, point_num, cost_function, pt_dist, start_state, end_state, end_n=3, dim=3,
                      knot_span=0.0, time_lb=-1.0, guide_pts=None, waypoints=None, waypt_idx=None, view=None,
                      **cfgkw):
    """BsplineOptimizer::combineCost of the reference; same signature as fuel_oracle.bspline_cost_grad."""
    L = lib()
    p = dict(fo.DEFAULT_BSPLINE)
    p.update(cfgkw)
    cfg = fo.BsplineCfg(**p)
    x = np.ascontiguousarray(x, dtype=np.float64)
    keep = []

    def ptr(a, dt=np.float64):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_double if dt == np.float64 else C.c_int))
    pb = fo.BsplineProblem()
    pb.cost_function, pb.dim, pb.point_num = cost_function, dim, point_num
    pb.knot_span, pb.pt_dist, pb.time_lb = knot_span, pt_dist, time_lb
    pb.start_state, pb.end_state, pb.end_n = ptr(start_state), ptr(end_state), end_n
    pb.guide_pts, pb.waypoints = ptr(guide_pts), ptr(waypoints)
    pb.waypt_idx = ptr(waypt_idx, np.int32)
    pb.n_waypt = 0 if waypoints is None else len(waypoints)
    if view is not None:
        pb.view_pt, pb.view_dir, pb.view_idx = ptr(view[0]), ptr(view[1]), int(view[2])
    cost = C.c_double()
    grad = np.zeros(len(x))
    L.ref_bspline_cost_grad(rmap.h, C.byref(cfg), C.byref(pb), fo._dp(x), C.byref(cost), fo._dp(grad))
    return cost.value, grad


# ---- MapROS::proessDepthImage from the real map_ros.cpp (separate shared object) ------------------
SO_MAPROS = os.path.join(os.path.dirname(_HERE), "_ref", "libfuel_ref_mapros.so")
_LIB_MR = None


def mapros_available():
    return os.path.exists(SO_MAPROS)


def project_depth(img, pos, quat_wxyz, cfg=None):
    """The reference's own proessDepthImage on a 16UC1 image: float32 [n,3]."""
    global _LIB_MR
    if _LIB_MR is None:
        _LIB_MR = C.CDLL(SO_MAPROS)
        _LIB_MR.ref_process_depth.restype = C.c_int
        _LIB_MR.ref_process_depth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.c_void_p, C.c_int]

    class RefDepthCfg(C.Structure):  # field order of ref_depth_api.cpp
        _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                    ("maxdist", C.c_double), ("mindist", C.c_double), ("scaling", C.c_double),
                    ("margin", C.c_int), ("skip", C.c_int)]
    c = cfg or fo.depth_cfg()
    rc = RefDepthCfg(c.fx, c.fy, c.cx, c.cy, c.depth_filter_maxdist, c.depth_filter_mindist,
                     c.k_depth_scaling_factor, c.depth_filter_margin, c.skip_pixel)
    img = np.ascontiguousarray(img, dtype=np.uint16)
    cap = img.shape[0] * img.shape[1]
    out = np.empty((cap, 3), dtype=np.float32)
    n = _LIB_MR.ref_process_depth(img.ctypes.data, img.shape[0], img.shape[1], C.byref(rc),
                                  (C.c_double * 3)(*[float(v) for v in pos]),
                                  (C.c_double * 4)(*[float(v) for v in quat_wxyz]), out.ctypes.data, cap)
    return out[:n].copy()


def spline_parameterize(ts, points, derivs, degree=3):
    """The real NonUniformBspline::parameterizeToBspline."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    derivs = np.ascontiguousarray(derivs, dtype=np.float64)
    L = lib()
    D = C.POINTER(C.c_double)
    L.ref_spline_parameterize.restype = None
    L.ref_spline_parameterize.argtypes = [C.c_double, D, C.c_int, D, C.c_int, D]
    ctrl = np.zeros((len(points) + degree - 1, 3))
    L.ref_spline_parameterize(float(ts), points.ctypes.data_as(D), len(points), derivs.ctypes.data_as(D), int(degree),
                              ctrl.ctypes.data_as(D))
    return ctrl


def spline_boundary_states(ctrl, ts, degree=3, ks=2, ke=0):
    """The real NonUniformBspline::getBoundaryStates on setUniformBspline(ctrl, degree, ts)."""
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    L = lib()
    D = C.POINTER(C.c_double)
    L.ref_spline_boundary_states.restype = None
    L.ref_spline_boundary_states.argtypes = [D, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, D, D]
    start = np.zeros((ks + 1, 3))
    end = np.zeros((ke + 1, 3))
    L.ref_spline_boundary_states(ctrl.ctypes.data_as(D), len(ctrl), int(degree), float(ts), int(ks), int(ke),
                                 start.ctypes.data_as(D), end.ctypes.data_as(D))
    return start, end
