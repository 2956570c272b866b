"""ctypes binding of the CPU oracle (oracle/libfuel_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never from fuel_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MapCfg(C.Structure):
    """Field-for-field the ROS parameters of sdf_map.cpp:19-47,78-82."""
    _fields_ = [
        ("resolution", C.c_double),
        ("map_size", C.c_double * 3),
        ("ground_height", C.c_double),
        ("obstacles_inflation", C.c_double),
        ("local_bound_inflate", C.c_double),
        ("default_dist", C.c_double),
        ("optimistic", C.c_int),
        ("signed_dist", C.c_int),
        ("p_hit", C.c_double),
        ("p_miss", C.c_double),
        ("p_min", C.c_double),
        ("p_max", C.c_double),
        ("p_occ", C.c_double),
        ("max_ray_length", C.c_double),
        ("virtual_ceil_height", C.c_double),
        ("box_min", C.c_double * 3),
        ("box_max", C.c_double * 3),
    ]


class FrontierCfg(C.Structure):
    _fields_ = [("cluster_min", C.c_int), ("min_z", C.c_double), ("cluster_size_xy", C.c_double),
                ("down_sample", C.c_int), ("split", C.c_int), ("canonical_order", C.c_int),
                ("flip_principal_dir", C.c_int)]


class BsplineCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("ld_smooth", "ld_dist", "ld_feasi", "ld_start", "ld_end", "ld_guide", "ld_waypt",
                 "ld_view", "ld_time", "dist0", "max_vel", "max_acc", "wnl", "dlmin")] + \
               [("bspline_degree", C.c_int)]


class BsplineProblem(C.Structure):
    _fields_ = [
        ("cost_function", C.c_int), ("dim", C.c_int), ("point_num", C.c_int),
        ("knot_span", C.c_double), ("pt_dist", C.c_double), ("time_lb", C.c_double),
        ("start_state", C.POINTER(C.c_double)), ("end_state", C.POINTER(C.c_double)),
        ("end_n", C.c_int),
        ("guide_pts", C.POINTER(C.c_double)), ("waypoints", C.POINTER(C.c_double)),
        ("waypt_idx", C.POINTER(C.c_int)), ("n_waypt", C.c_int),
        ("view_pt", C.POINTER(C.c_double)), ("view_dir", C.POINTER(C.c_double)),
        ("view_idx", C.c_int),
    ]


class ViewpointCfg(C.Structure):
    """frontier/candidate_* + perception_utils/* (algorithm.xml:106-121)."""
    _fields_ = [("candidate_rmin", C.c_double), ("candidate_rmax", C.c_double), ("candidate_rnum", C.c_int),
                ("candidate_dphi", C.c_double), ("min_candidate_clearance", C.c_double),
                ("min_visib_num", C.c_int), ("min_candidate_dist", C.c_double),
                ("min_view_finish_fraction", C.c_double), ("top_angle", C.c_double), ("left_angle", C.c_double),
                ("right_angle", C.c_double), ("max_dist", C.c_double)]


def viewpoint_cfg(rmin=1.5, rmax=2.5, rnum=3, dphi=15 * 3.1415926 / 180.0, clearance=0.21, min_visib_num=15,
                  min_candidate_dist=0.75, min_view_finish_fraction=0.2, top_angle=0.56125, left_angle=0.69222,
                  right_angle=0.68901, max_dist=4.5, cls=None):
    return (cls or ViewpointCfg)(rmin, rmax, rnum, dphi, clearance, min_visib_num, min_candidate_dist,
                                 min_view_finish_fraction, top_angle, left_angle, right_angle, max_dist)


class DepthCfg(C.Structure):
    _fields_ = [
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("depth_filter_maxdist", C.c_double), ("depth_filter_mindist", C.c_double),
        ("depth_filter_margin", C.c_int),
        ("k_depth_scaling_factor", C.c_double),
        ("skip_pixel", C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libfuel_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("fuel_oracle.cpp", "fuel_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        P = C.c_void_p
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.fo_map_create.restype = P
        L.fo_map_create.argtypes = [C.POINTER(MapCfg)]
        L.fo_map_destroy.argtypes = [P]
        for name in ("fo_map_occupancy", "fo_map_distance", "fo_map_distance_neg"):
            getattr(L, name).restype = dp
            getattr(L, name).argtypes = [P]
        for name in ("fo_map_inflate", "fo_map_flag_rayend"):
            getattr(L, name).restype = C.POINTER(C.c_char)
            getattr(L, name).argtypes = [P]
        L.fo_map_voxel_num.argtypes = [P, ip]
        L.fo_map_origin.argtypes = [P, dp]
        L.fo_map_box_index.argtypes = [P, ip, ip]
        L.fo_map_logodds.argtypes = [P, dp]
        L.fo_map_input_points.argtypes = [P, C.c_void_p, C.c_int, C.c_int, dp]
        L.fo_map_inflate_local.argtypes = [P]
        L.fo_map_update_esdf.argtypes = [P]
        L.fo_map_reset_buffer_all.argtypes = [P]
        L.fo_map_reset_buffer.argtypes = [P, dp, dp]
        L.fo_map_set_occupied.argtypes = [P, dp, C.c_int]
        L.fo_map_get_local_bound.argtypes = [P, ip, ip]
        L.fo_map_set_local_bound.argtypes = [P, ip, ip]
        L.fo_map_get_updated_box.argtypes = [P, dp, dp, C.c_int]
        L.fo_map_set_updated_box.argtypes = [P, dp, dp]
        L.fo_map_get_occupancy_idx.argtypes = [P, ip]
        L.fo_map_get_occupancy_pos.argtypes = [P, dp]
        L.fo_map_get_inflate_idx.argtypes = [P, ip]
        L.fo_map_get_distance_idx.restype = C.c_double
        L.fo_map_get_distance_idx.argtypes = [P, ip]
        L.fo_map_dist_grad.argtypes = [P, dp, C.c_int, dp, dp]
        L.fo_raycast_cells.restype = C.c_int
        L.fo_raycast_cells.argtypes = [P, dp, dp, ip, C.c_int]
        L.fo_frontier_create.restype = P
        L.fo_frontier_create.argtypes = [P, C.POINTER(FrontierCfg)]
        L.fo_frontier_destroy.argtypes = [P]
        L.fo_frontier_flags.restype = C.POINTER(C.c_char)
        L.fo_frontier_flags.argtypes = [P]
        L.fo_frontier_search.argtypes = [P]
        L.fo_frontier_commit.argtypes = [P, C.c_int]
        L.fo_frontier_count.argtypes = [P, C.c_int]
        L.fo_frontier_cluster_size.argtypes = [P, C.c_int, C.c_int]
        L.fo_frontier_cluster_cells.argtypes = [P, C.c_int, C.c_int, ip]
        L.fo_frontier_cluster_info.argtypes = [P, C.c_int, C.c_int, dp]
        L.fo_frontier_set_viewpoint_cfg.argtypes = [P, C.POINTER(ViewpointCfg)]
        L.fo_frontier_compute_to_visit.argtypes = [P]
        L.fo_frontier_is_covered.argtypes = [P]
        L.fo_frontier_viewpoint_count.argtypes = [P, C.c_int, C.c_int]
        L.fo_frontier_viewpoints.argtypes = [P, C.c_int, C.c_int, dp, ip]
        L.fo_frontier_cluster_filtered_size.argtypes = [P, C.c_int, C.c_int]
        L.fo_frontier_cluster_filtered.argtypes = [P, C.c_int, C.c_int, dp]
        L.fo_frontier_removed_count.argtypes = [P]
        L.fo_frontier_removed_ids.argtypes = [P, ip]
        L.fo_bspline_pt_dist.restype = C.c_double
        L.fo_bspline_pt_dist.argtypes = [dp, C.c_int, C.c_int]
        L.fo_bspline_cost_grad.argtypes = [P, C.POINTER(BsplineCfg), C.POINTER(BsplineProblem), dp, dp, dp]
        L.fo_project_depth.restype = C.c_int
        L.fo_project_depth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(DepthCfg), dp, dp, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def depth_cfg(fx=387.229248046875, fy=387.229248046875, cx=321.04638671875, cy=243.44969177246094,
              maxdist=5.0, mindist=0.2, margin=2, scaling=1000.0, skip=2):
    return DepthCfg(fx, fy, cx, cy, maxdist, mindist, margin, scaling, skip)


def project_depth(img, pos, quat_wxyz, cfg=None):
    """MapROS::proessDepthImage restated (map_ros.cpp:176-215): float32 [n,3] world points."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    cfg = cfg or depth_cfg()
    cap = img.shape[0] * img.shape[1]
    out = np.empty((cap, 3), dtype=np.float32)
    n = lib().fo_project_depth(img.ctypes.data, img.shape[0], img.shape[1], C.byref(cfg),
                               (C.c_double * 3)(*[float(v) for v in pos]),
                               (C.c_double * 4)(*[float(v) for v in quat_wxyz]), out.ctypes.data, cap)
    return out[:n].copy()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


# exploration.launch / algorithm.xml values
DEFAULT_MAP = dict(resolution=0.1, ground_height=-1.0, obstacles_inflation=0.199,
                   local_bound_inflate=0.5, default_dist=0.0, optimistic=0, signed_dist=0,
                   p_hit=0.65, p_miss=0.35, p_min=0.12, p_max=0.90, p_occ=0.80,
                   max_ray_length=4.5, virtual_ceil_height=-10.0)
DEFAULT_BSPLINE = dict(ld_smooth=20.0, ld_dist=10.0, ld_feasi=2.0, ld_start=100.0, ld_end=0.5,
                       ld_guide=1.5, ld_waypt=0.3, ld_view=0.0, ld_time=1.0, dist0=0.7,
                       max_vel=2.0, max_acc=2.0, wnl=1.0, dlmin=0.0, bspline_degree=3)


def make_cfg(cls, map_size, box_min=None, box_max=None, **kw):
    """Build a MapCfg-like ctypes struct (works for the oracle's and the product's struct)."""
    p = dict(DEFAULT_MAP)
    p.update(kw)
    c = cls()
    for k, v in p.items():
        if hasattr(c, k):
            setattr(c, k, v)
    for i in range(3):
        c.map_size[i] = float(map_size[i])
    org = (-map_size[0] / 2.0, -map_size[1] / 2.0, p["ground_height"])
    bmin = box_min if box_min is not None else org
    bmax = box_max if box_max is not None else tuple(org[i] + map_size[i] for i in range(3))
    for i in range(3):
        c.box_min[i] = float(bmin[i])
        c.box_max[i] = float(bmax[i])
    return c


class OracleMap:
    def __init__(self, map_size, box_min=None, box_max=None, **kw):
        self.L = lib()
        self.cfg = make_cfg(MapCfg, map_size, box_min, box_max, **kw)
        self.h = self.L.fo_map_create(C.byref(self.cfg))
        nv = (C.c_int * 3)()
        self.L.fo_map_voxel_num(self.h, nv)
        self.nvox = tuple(nv)
        self.N = self.nvox[0] * self.nvox[1] * self.nvox[2]
        o = (C.c_double * 3)()
        self.L.fo_map_origin(self.h, o)
        self.origin = np.array(o)
        self.res = self.cfg.resolution
        lo = (C.c_double * 5)()
        self.L.fo_map_logodds(self.h, lo)
        self.l_hit, self.l_miss, self.l_min, self.l_max, self.l_occ = list(lo)
        self.occ = np.ctypeslib.as_array(self.L.fo_map_occupancy(self.h), shape=(self.N,))
        self.dist = np.ctypeslib.as_array(self.L.fo_map_distance(self.h), shape=(self.N,))
        self.dist_neg = np.ctypeslib.as_array(self.L.fo_map_distance_neg(self.h), shape=(self.N,))
        self.infl = np.ctypeslib.as_array(
            C.cast(self.L.fo_map_inflate(self.h), C.POINTER(C.c_int8)), shape=(self.N,))
        self.flag_rayend = np.ctypeslib.as_array(
            C.cast(self.L.fo_map_flag_rayend(self.h), C.POINTER(C.c_int8)), shape=(self.N,))

    def __del__(self):
        try:
            self.L.fo_map_destroy(self.h)
        except Exception:
            pass

    @property
    def unknown_value(self):
        return self.l_min - 0.01

    def box_index(self):
        a, b = (C.c_int * 3)(), (C.c_int * 3)()
        self.L.fo_map_box_index(self.h, a, b)
        return tuple(a), tuple(b)

    def input_points(self, pts, cam):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        self.L.fo_map_input_points(self.h, pts.ctypes.data, 12, len(pts), _d3(cam))

    def inflate_local(self):
        self.L.fo_map_inflate_local(self.h)

    def update_esdf(self):
        self.L.fo_map_update_esdf(self.h)

    def reset_buffer(self, lo=None, hi=None):
        if lo is None:
            self.L.fo_map_reset_buffer_all(self.h)
        else:
            self.L.fo_map_reset_buffer(self.h, _d3(lo), _d3(hi))

    def set_occupied(self, pos, occ=1):
        self.L.fo_map_set_occupied(self.h, _d3(pos), occ)

    def get_local_bound(self):
        a, b = (C.c_int * 3)(), (C.c_int * 3)()
        self.L.fo_map_get_local_bound(self.h, a, b)
        return tuple(a), tuple(b)

    def set_local_bound(self, lo, hi):
        self.L.fo_map_set_local_bound(self.h, _i3(lo), _i3(hi))

    def get_updated_box(self, reset=False):
        a, b = (C.c_double * 3)(), (C.c_double * 3)()
        self.L.fo_map_get_updated_box(self.h, a, b, int(reset))
        return np.array(a), np.array(b)

    def set_updated_box(self, lo, hi):
        self.L.fo_map_set_updated_box(self.h, _d3(lo), _d3(hi))

    def dist_grad(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        d = np.empty(len(pos))
        g = np.empty((len(pos), 3))
        self.L.fo_map_dist_grad(self.h, _dp(pos), len(pos), _dp(d), _dp(g))
        return d, g

    def raycast_cells(self, start, end, cap=4096):
        out = np.empty((cap, 3), dtype=np.int32)
        n = self.L.fo_raycast_cells(self.h, _d3(start), _d3(end), _ip(out), cap)
        return out[:n].copy()

    # ---- synthetic inputs (generator lives in fuel_amd/synth, independent of the oracle) ----
    def synth(self):
        from fuel_amd import synth
        return synth.World(self.nvox, self.origin, self.res,
                           [self.l_hit, self.l_miss, self.l_min, self.l_max, self.l_occ])

    def fixture_world(self, seed, n_obstacles):
        return self.synth().world(seed, n_obstacles)

    def fixture_known_state(self, truth, seed, n_spheres, rmin=3.0, rmax=4.5):
        """Overwrite occupancy_buffer_ with the as-if-explored state; returns #known voxels."""
        return self.synth().known_state(truth, seed, n_spheres, rmin, rmax, out=self.occ)[1]

    def fixture_camera(self, truth, seed, k, n_total, extent_frac=0.8):
        return self.synth().camera(truth, seed, k, n_total, extent_frac)

    def fixture_render(self, truth, pose, width=640, height=480, skip=2, margin=2, maxdist=5.0, mindist=0.2):
        return self.synth().render(truth, pose, width, height, skip, margin, maxdist, mindist)


class OracleFrontier:
    def __init__(self, omap, cluster_min=100, min_z=0.4, cluster_size_xy=2.0, down_sample=0, split=False,
                 canonical_order=False, flip_principal_dir=False):
        """down_sample=3 fills filtered_cells_; split=True runs splitLargeFrontiers (needs down_sample);
        canonical_order=True feeds the VoxelGrid the cells in ascending voxel address instead of BFS order
        (libfuelmi's cell order: only the float summation order inside a leaf changes)."""
        self.L = lib()
        self.map = omap
        if split and down_sample <= 0:
            down_sample = 3
        cfg = FrontierCfg(cluster_min, min_z, cluster_size_xy, down_sample, int(split), int(canonical_order),
                          int(flip_principal_dir))
        self.h = self.L.fo_frontier_create(omap.h, C.byref(cfg))
        self.flags = np.ctypeslib.as_array(
            C.cast(self.L.fo_frontier_flags(self.h), C.POINTER(C.c_int8)), shape=(omap.N,))

    def __del__(self):
        try:
            self.L.fo_frontier_destroy(self.h)
        except Exception:
            pass

    def search(self):
        return self.L.fo_frontier_search(self.h)

    def commit(self, dormant=False):
        self.L.fo_frontier_commit(self.h, int(dormant))

    def clusters(self, which=0):
        out = []
        for k in range(self.L.fo_frontier_count(self.h, which)):
            n = self.L.fo_frontier_cluster_size(self.h, which, k)
            a = np.empty(n, dtype=np.int32)
            self.L.fo_frontier_cluster_cells(self.h, which, k, _ip(a))
            out.append(a)
        return out

    def cluster_info(self, which, k):
        o = np.empty(9)
        self.L.fo_frontier_cluster_info(self.h, which, k, _dp(o))
        return o[:3], o[3:6], o[6:9]

    def set_viewpoint_cfg(self, cfg):
        self.L.fo_frontier_set_viewpoint_cfg(self.h, C.byref(cfg))

    def compute_to_visit(self):
        """computeFrontiersToVisit: viewpoints for tmp clusters, then frontiers_ / dormant_frontiers_."""
        self.L.fo_frontier_compute_to_visit(self.h)

    def is_covered(self):
        return bool(self.L.fo_frontier_is_covered(self.h))

    def viewpoints(self, which, k):
        """(pos_yaw float64 [n,4], visib_num int32 [n]) of cluster k, best coverage first."""
        n = self.L.fo_frontier_viewpoint_count(self.h, which, k)
        py = np.empty((n, 4))
        vis = np.empty(n, dtype=np.int32)
        if n:
            self.L.fo_frontier_viewpoints(self.h, which, k, _dp(py), _ip(vis))
        return py, vis

    def filtered(self, which, k):
        """Frontier::filtered_cells_ of cluster k: float64 [n,3] (values carry float32 precision)."""
        n = self.L.fo_frontier_cluster_filtered_size(self.h, which, k)
        o = np.empty((n, 3))
        if n:
            self.L.fo_frontier_cluster_filtered(self.h, which, k, _dp(o))
        return o

    def removed_ids(self):
        n = self.L.fo_frontier_removed_count(self.h)
        a = np.empty(n, dtype=np.int32)
        if n:
            self.L.fo_frontier_removed_ids(self.h, _ip(a))
        return a


COST = dict(SMOOTHNESS=1, DISTANCE=2, FEASIBILITY=4, START=8, END=16, GUIDE=32, WAYPOINTS=64,
            VIEWCONS=128, MINTIME=256)
COST["GUIDE_PHASE"] = COST["SMOOTHNESS"] | COST["GUIDE"] | COST["START"] | COST["END"]
COST["NORMAL_PHASE"] = (COST["SMOOTHNESS"] | COST["DISTANCE"] | COST["FEASIBILITY"] | COST["START"]
                        | COST["END"])


def bspline_pt_dist(ctrl):
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    n, dim = ctrl.shape
    return lib().fo_bspline_pt_dist(_dp(ctrl), n, dim)


def spline_parameterize(ts, points, derivs, degree=3):
    """parameterizeToBspline (non_uniform_bspline.cpp:178-265): points [K][3], derivs [4][3] -> ctrl [K+degree-1][3]."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    derivs = np.ascontiguousarray(derivs, dtype=np.float64)
    L = lib()
    L.fo_spline_parameterize.restype = C.c_int
    L.fo_spline_parameterize.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                         C.POINTER(C.c_double)]
    ctrl = np.zeros((len(points) + degree - 1, 3))
    if L.fo_spline_parameterize(float(ts), _dp(points), len(points), _dp(derivs), int(degree), _dp(ctrl)):
        raise ValueError("parameterizeToBspline: refused inputs")
    return ctrl


def spline_boundary_states(ctrl, ts, degree=3, ks=2, ke=0):
    """getBoundaryStates(ks, ke) (non_uniform_bspline.cpp:107-122) -> (start [ks+1][3], end [ke+1][3])."""
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    L = lib()
    L.fo_spline_boundary_states.restype = None
    L.fo_spline_boundary_states.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]
    start = np.zeros((ks + 1, 3))
    end = np.zeros((ke + 1, 3))
    L.fo_spline_boundary_states(_dp(ctrl), len(ctrl), int(degree), float(ts), int(ks), int(ke), _dp(start), _dp(end))
    return start, end


def _bspline_setup(x, point_num, cost_function, pt_dist, start_state, end_state, end_n=3,
                      dim=3, knot_span=0.0, time_lb=-1.0, guide_pts=None, waypoints=None,
                      waypt_idx=None, view=None, **cfgkw):
    """(lib, cfg, problem, x, keep-alive list) shared by the evaluation and the optimiser."""
    L = lib()
    p = dict(DEFAULT_BSPLINE)
    p.update(cfgkw)
    cfg = BsplineCfg(**p)
    x = np.ascontiguousarray(x, dtype=np.float64)
    keep = []

    def ptr(a, dt=np.float64):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_double if dt == np.float64 else C.c_int))
    pb = BsplineProblem()
    pb.cost_function = cost_function
    pb.dim = dim
    pb.point_num = point_num
    pb.knot_span = knot_span
    pb.pt_dist = pt_dist
    pb.time_lb = time_lb
    pb.start_state = ptr(start_state)
    pb.end_state = ptr(end_state)
    pb.end_n = end_n
    pb.guide_pts = ptr(guide_pts)
    pb.waypoints = ptr(waypoints)
    pb.waypt_idx = ptr(waypt_idx, np.int32)
    pb.n_waypt = 0 if waypoints is None else len(waypoints)
    if view is not None:
        pb.view_pt = ptr(view[0])
        pb.view_dir = ptr(view[1])
        pb.view_idx = int(view[2])
    return L, cfg, pb, x, keep


def bspline_cost_grad(omap, x, point_num, cost_function, pt_dist, start_state, end_state, end_n=3,
                      dim=3, knot_span=0.0, time_lb=-1.0, guide_pts=None, waypoints=None,
                      waypt_idx=None, view=None, **cfgkw):
    """One combineCost evaluation.  Returns (cost, grad)."""
    L, cfg, pb, x, keep = _bspline_setup(x, point_num, cost_function, pt_dist, start_state, end_state, end_n, dim,
                                         knot_span, time_lb, guide_pts, waypoints, waypt_idx, view, **cfgkw)
    cost = C.c_double()
    grad = np.zeros(len(x))
    L.fo_bspline_cost_grad(omap.h, C.byref(cfg), C.byref(pb), _dp(x), C.byref(cost), _dp(grad))
    return cost.value, grad


def bspline_optimize(omap, x, point_num, cost_function, pt_dist, start_state, end_state, end_n=3, dim=3,
                     knot_span=0.0, time_lb=-1.0, guide_pts=None, waypoints=None, waypt_idx=None, view=None,
                     max_eval=300, **cfgkw):
    """BsplineOptimizer::optimize() with the in-house L-BFGS: returns (best_x, best_cost, evaluations)."""
    L, cfg, pb, x, keep = _bspline_setup(x, point_num, cost_function, pt_dist, start_state, end_state, end_n, dim,
                                         knot_span, time_lb, guide_pts, waypoints, waypt_idx, view, **cfgkw)
    L.fo_bspline_optimize.restype = C.c_double
    L.fo_bspline_optimize.argtypes = [C.c_void_p, C.POINTER(BsplineCfg), C.POINTER(BsplineProblem),
                                      C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    xo = x.copy()
    ev = C.c_int()
    f = L.fo_bspline_optimize(omap.h, C.byref(cfg), C.byref(pb), _dp(xo), int(max_eval), C.byref(ev))
    return xo, f, ev.value
