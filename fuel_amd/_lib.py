"""ctypes loader for libfuelmi.so (the HIP/gfx950 implementation behind include/fuelmi.h).

There is NO CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"` (hipcc, in-tree).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FUELMI_LIB_PATH") or os.path.join(_HERE, "libfuelmi.so")  # (override: A/B runs of two builds)
_LIB = None

K_INFLATE, K_ESDF_ZY, K_ESDF_X, K_FRONTIER, K_BSPLINE, K_INSERT, K_COUNT = range(7)


class FuelmiError(RuntimeError):
    pass


class MapCfg(C.Structure):
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
        ("device", C.c_int),
    ]


class DepthCfg(C.Structure):
    """map_ros/... parameters of MapROS (plan_env/src/map_ros.cpp:22-30); defaults = exploration.launch / algorithm.xml."""
    _fields_ = [
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("depth_filter_maxdist", C.c_double), ("depth_filter_mindist", C.c_double),
        ("depth_filter_margin", C.c_int),
        ("k_depth_scaling_factor", C.c_double),
        ("skip_pixel", C.c_int),
    ]


class MapInfo(C.Structure):
    _fields_ = [
        ("voxel_num", C.c_int * 3),
        ("origin", C.c_double * 3),
        ("min_boundary", C.c_double * 3),
        ("max_boundary", C.c_double * 3),
        ("resolution_inv", C.c_double),
        ("box_min", C.c_int * 3),
        ("box_max", C.c_int * 3),
        ("prob_hit_log", C.c_double),
        ("prob_miss_log", C.c_double),
        ("clamp_min_log", C.c_double),
        ("clamp_max_log", C.c_double),
        ("min_occupancy_log", C.c_double),
        ("inflate_step", C.c_int),
    ]


class FrontierCfg(C.Structure):
    _fields_ = [("cluster_min", C.c_int), ("min_z", C.c_double), ("cluster_size_xy", C.c_double),
                ("down_sample", C.c_int), ("split", C.c_int), ("reference_order", C.c_int)]


class ViewpointCfg(C.Structure):
    """frontier/candidate_* + perception_utils/* parameters (algorithm.xml:106-121)."""
    _fields_ = [("candidate_rmin", C.c_double), ("candidate_rmax", C.c_double), ("candidate_rnum", C.c_int),
                ("candidate_dphi", C.c_double), ("min_candidate_clearance", C.c_double),
                ("min_visib_num", C.c_int), ("min_candidate_dist", C.c_double),
                ("min_view_finish_fraction", C.c_double), ("top_angle", C.c_double), ("left_angle", C.c_double),
                ("right_angle", C.c_double), ("max_dist", C.c_double)]


class BsplineCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("ld_smooth", "ld_dist", "ld_feasi", "ld_start", "ld_end", "ld_guide", "ld_waypt",
                 "ld_view", "ld_time", "dist0", "max_vel", "max_acc", "wnl", "dlmin")] + \
               [("bspline_degree", C.c_int)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class BsplineBatch(C.Structure):
    _fields_ = [
        ("cost_function", C.c_int), ("dim", C.c_int), ("point_num", C.c_int), ("n_traj", C.c_int),
        ("x", _dp), ("pt_dist", _dp), ("knot_span", _dp), ("time_lb", _dp),
        ("start_state", _dp), ("end_state", _dp), ("end_n", C.c_int),
        ("guide_pts", _dp), ("waypoints", _dp), ("waypt_idx", _ip), ("n_waypt", C.c_int),
        ("view_pt", _dp), ("view_dir", _dp), ("view_idx", _ip),
    ]


# every symbol include/fuelmi.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    "fuelmi_last_error": (C.c_char_p, []),
    "fuelmi_version": (C.c_char_p, []),
    "fuelmi_init": (C.c_int, [C.c_int]),
    "fuelmi_hw_queues": (C.c_int, []),
    "fuelmi_hw_queues_state": (C.c_int, []),
    "fuelmi_device_count": (C.c_int, []),
    "fuelmi_map_create": (C.c_int, [C.POINTER(MapCfg), _PP]),
    "fuelmi_map_destroy": (None, [_P]),
    "fuelmi_map_get_info": (C.c_int, [_P, C.POINTER(MapInfo)]),
    "fuelmi_map_input_points": (C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, _dp]),
    "fuelmi_map_input_depth": (C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.POINTER(DepthCfg), _dp, _dp,
                                         C.POINTER(C.c_int)]),
    "fuelmi_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "fuelmi_host_unregister": (C.c_int, [C.c_void_p]),
    "fuelmi_device_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "fuelmi_device_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "fuelmi_device_free": (C.c_int, [C.c_void_p]),
    "fuelmi_hbm_triad": (C.c_int, [C.c_int, C.c_size_t, C.c_int, _dp]),
    "fuelmi_hbm_expand": (C.c_int, [C.c_int, C.c_size_t, C.c_int, _dp]),
    "fuelmi_map_project_depth": (C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.POINTER(DepthCfg), _dp, _dp,
                                           C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "fuelmi_map_inflate_local": (C.c_int, [_P]),
    "fuelmi_map_update_esdf": (C.c_int, [_P]),
    "fuelmi_map_set_esdf_family": (C.c_int, [_P, C.c_int]),
    "fuelmi_map_last_esdf_family": (C.c_int, [_P]),
    "fuelmi_map_last_inflate_kernel": (C.c_int, [_P]),
    "fuelmi_map_reset_buffer_all": (C.c_int, [_P]),
    "fuelmi_map_reset_buffer": (C.c_int, [_P, _dp, _dp]),
    "fuelmi_map_set_occupied": (C.c_int, [_P, _dp, C.c_int, C.c_int]),
    "fuelmi_map_get_local_bound": (C.c_int, [_P, _ip, _ip]),
    "fuelmi_map_set_local_bound": (C.c_int, [_P, _ip, _ip]),
    "fuelmi_map_get_updated_box": (C.c_int, [_P, _dp, _dp, C.c_int]),
    "fuelmi_map_set_updated_box": (C.c_int, [_P, _dp, _dp]),
    "fuelmi_map_upload_occupancy": (C.c_int, [_P, _dp]),
    "fuelmi_map_sync_host": (C.c_int, [_P, _ip, _ip, _dp, C.c_void_p, _dp]),
    "fuelmi_map_register_mirrors": (C.c_int, [_P, _dp, C.c_void_p, _dp]),
    "fuelmi_map_unregister_mirrors": (C.c_int, [_P]),
    "fuelmi_map_dist_grad": (C.c_int, [_P, _dp, C.c_int, _dp, _dp]),
    "fuelmi_map_coarse_dist": (C.c_int, [_P, _dp, C.c_int, _dp]),
    "fuelmi_map_query_state": (C.c_int, [_P, _ip, C.c_int, _ip, _ip]),
    "fuelmi_map_synchronize": (C.c_int, [_P]),
    "fuelmi_frontier_set_viewpoint_cfg": (C.c_int, [_P, C.POINTER(ViewpointCfg)]),
    "fuelmi_frontier_compute_to_visit": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fuelmi_frontier_viewpoint_count": (C.c_int, [_P, C.c_int, C.c_int]),
    "fuelmi_frontier_viewpoints": (C.c_int, [_P, C.c_int, C.c_int, _dp, C.POINTER(C.c_int)]),
    "fuelmi_frontier_is_covered": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "fuelmi_frontier_cluster_filtered_size": (C.c_int, [_P, C.c_int, C.c_int]),
    "fuelmi_frontier_cluster_filtered": (C.c_int, [_P, C.c_int, C.c_int, C.c_void_p]),
    "fuelmi_frontier_create": (C.c_int, [_P, C.POINTER(FrontierCfg), _PP]),
    "fuelmi_frontier_destroy": (None, [_P]),
    "fuelmi_frontier_reset": (C.c_int, [_P]),
    "fuelmi_frontier_stats": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "fuelmi_frontier_resolved_in_launch": (C.c_int, [_P]),
    "fuelmi_frontier_order_stats": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "fuelmi_frontier_synchronize": (C.c_int, [_P]),
    "fuelmi_bench_cycles": (C.c_int, [_P, _P, _P, _dp, _dp, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]),
    "fuelmi_bench_host_profile": (C.c_int, [_P, _dp]),
    "fuelmi_bench_cycles_delivered": (C.c_int, [_P, _P, _P, _dp, _dp, C.c_int, _ip, C.c_size_t, _dp, _dp,
                                                C.POINTER(C.c_int), _dp]),
    "fuelmi_bench_stream": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(DepthCfg),
                                      _dp, _dp, C.c_int, C.POINTER(C.c_int), _dp, _dp]),
    "fuelmi_frontier_search": (C.c_int, [_P, _ip]),
    "fuelmi_frontier_search_begin": (C.c_int, [_P]),
    "fuelmi_frontier_search_end": (C.c_int, [_P, _ip]),
    "fuelmi_frontier_keep_previous": (C.c_int, [_P, C.c_int]),
    "fuelmi_frontier_commit": (C.c_int, [_P, C.c_int]),
    "fuelmi_frontier_count": (C.c_int, [_P, C.c_int]),
    "fuelmi_frontier_cluster_size": (C.c_int, [_P, C.c_int, C.c_int]),
    "fuelmi_frontier_cluster_cells": (C.c_int, [_P, C.c_int, C.c_int, _ip]),
    "fuelmi_frontier_cluster_centres": (C.c_int, [_P, C.c_int, C.c_int, _dp]),
    "fuelmi_frontier_cluster_info": (C.c_int, [_P, C.c_int, C.c_int, _dp]),
    "fuelmi_frontier_removed_count": (C.c_int, [_P]),
    "fuelmi_frontier_removed_ids": (C.c_int, [_P, _ip]),
    "fuelmi_frontier_get_flags": (C.c_int, [_P, C.c_void_p]),
    "fuelmi_bspline_cost_grad": (C.c_int, [_P, C.POINTER(BsplineCfg), C.POINTER(BsplineBatch), _dp, _dp]),
    "fuelmi_bspline_optimize": (C.c_int, [_P, C.POINTER(BsplineCfg), C.POINTER(BsplineBatch), C.c_int, C.c_double, _dp, _dp,
                                C.POINTER(C.c_int)]),
    "fuelmi_bspline_dev_create": (C.c_int, [_P, C.POINTER(BsplineCfg), C.POINTER(BsplineBatch), _PP]),
    "fuelmi_bspline_dev_eval": (C.c_int, [_P]),
    "fuelmi_bspline_dev_eval_pinned": (C.c_int, [_P, C.c_int]),
    "fuelmi_bspline_dev_collect": (C.c_int, [_P, C.c_int, _dp, _dp]),
    "fuelmi_bspline_dev_download": (C.c_int, [_P, _dp, _dp]),
    "fuelmi_bspline_dev_optimize": (C.c_int, [_P, C.c_int, _dp, _dp, C.POINTER(C.c_int)]),
    "fuelmi_bspline_dev_optimize_timed": (C.c_int, [_P, C.c_int, C.c_double, _dp, _dp, C.POINTER(C.c_int)]),
    "fuelmi_bspline_dev_destroy": (None, [_P]),
    "fuelmi_bspline_parameterize": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
    "fuelmi_bspline_boundary_states": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp]),
    "fuelmi_bspline_dev_load_samples": (C.c_int, [_P, C.c_int, _dp, _dp, _dp]),
    "fuelmi_timer_begin": (C.c_int, [_P]),
    "fuelmi_timer_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "fuelmi_profile_enable": (C.c_int, [_P, C.c_uint]),
    "fuelmi_profile_get": (C.c_int, [_P, C.c_int, _ip, _dp]),
    "fuelmi_profile_get_samples": (C.c_int, [_P, C.c_int, _dp, C.c_int, _ip]),
    "fuelmi_profile_get_timeline": (C.c_int, [_P, C.c_int, _dp, _dp, C.c_int, _ip]),
}


def lib():
    """Load libfuelmi.so; raises FuelmiError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FuelmiError(
                "%s not found: build the HIP extension first (__graft_entry__.build()); "
                "fuel_amd has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        # explicit process set-up (include/fuelmi.h fuelmi_init): GPU_MAX_HW_QUEUES=16 unless the environment decides --
        # effective only if nothing in this process has initialised HIP yet (import fuel_amd before torch.cuda is used)
        L.fuelmi_init(0)
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().fuelmi_last_error()
        raise FuelmiError("libfuelmi error %d: %s" % (rc, msg.decode() if msg else "?"))
