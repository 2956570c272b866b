"""Synthetic input generator (host side): seeded worlds, as-if-explored occupancy states and
depth-frame point clouds for bench.py and the tests.  Independent of the oracle and of libfuelmi.

Built in-tree with g++ (no GPU code): fuel_amd/synth/libfuel_synth.so.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfuel_synth.so")
_SRC = os.path.join(_HERE, "synth.cpp")
_LIB = None

# pinhole intrinsics, exploration_manager/launch/exploration.launch:38-41
CAM = dict(fx=387.229248046875, fy=387.229248046875, cx=321.04638671875, cy=243.44969177246094)


class Grid(C.Structure):
    _fields_ = [("nv", C.c_int * 3), ("origin", C.c_double * 3), ("res", C.c_double),
                ("logodds", C.c_double * 5)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SRC) > os.path.getmtime(_SO):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                               "-o", _SO, _SRC])
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        G = C.POINTER(Grid)
        dp = C.POINTER(C.c_double)
        L.synth_world.restype = C.c_long
        L.synth_world.argtypes = [G, C.c_uint64, C.c_int, C.c_void_p]
        L.synth_known_state.restype = C.c_long
        L.synth_known_state.argtypes = [G, C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_double, dp]
        L.synth_camera.argtypes = [G, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_double, dp]
        L.synth_render.restype = C.c_int
        L.synth_render.argtypes = [G, C.c_void_p, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_void_p, C.c_int]
        L.synth_depth_image.restype = None
        L.synth_depth_image.argtypes = [G, C.c_void_p, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def logodds(p_hit=0.65, p_miss=0.35, p_min=0.12, p_max=0.90, p_occ=0.80):
    """logit() of the fusion probabilities, as SDFMap::initMap computes them (sdf_map.cpp:49-54)."""
    lg = lambda x: math.log(x / (1 - x))  # noqa: E731
    return [lg(p_hit), lg(p_miss), lg(p_min), lg(p_max), lg(p_occ)]


class World:
    """A voxel grid geometry plus its generated ground truth."""

    def __init__(self, nvox, origin, res, lo=None):
        self.L = lib()
        g = Grid()
        for i in range(3):
            g.nv[i] = int(nvox[i])
            g.origin[i] = float(origin[i])
        g.res = float(res)
        for i, v in enumerate(lo if lo is not None else logodds()):
            g.logodds[i] = v
        self.g = g
        self.nvox = tuple(int(v) for v in nvox)
        self.N = self.nvox[0] * self.nvox[1] * self.nvox[2]

    @classmethod
    def for_map_size(cls, map_size, res=0.1, ground_height=-1.0, lo=None):
        nv = [int(math.ceil(map_size[i] / res)) for i in range(3)]
        org = (-map_size[0] / 2.0, -map_size[1] / 2.0, ground_height)
        return cls(nv, org, res, lo)

    def world(self, seed, n_obstacles):
        truth = np.zeros(self.N, dtype=np.uint8)
        self.L.synth_world(C.byref(self.g), seed, n_obstacles, truth.ctypes.data)
        return truth

    def known_state(self, truth, seed, n_spheres, rmin=3.0, rmax=4.5, out=None):
        occ = out if out is not None else np.empty(self.N)
        n = self.L.synth_known_state(C.byref(self.g), truth.ctypes.data, seed, n_spheres, rmin, rmax,
                                     occ.ctypes.data_as(C.POINTER(C.c_double)))
        return occ, n

    def camera(self, truth, seed, k, n_total, extent_frac=0.8):
        pose = (C.c_double * 5)()
        self.L.synth_camera(C.byref(self.g), truth.ctypes.data, seed, k, n_total, extent_frac, pose)
        return np.array(pose)

    def render(self, truth, pose, width=640, height=480, skip=2, margin=2, maxdist=5.0, mindist=0.2):
        cap = ((height - 2 * margin + skip - 1) // skip) * ((width - 2 * margin + skip - 1) // skip)
        out = np.empty((cap, 3), dtype=np.float32)
        s = width / 640.0  # intrinsics scale with the image width: small frames keep the field of view
        n = self.L.synth_render(C.byref(self.g), truth.ctypes.data, (C.c_double * 5)(*pose), width, height,
                                skip, margin, CAM["fx"] * s, CAM["fy"] * s, CAM["cx"] * s, CAM["cy"] * s,
                                maxdist, mindist, out.ctypes.data, cap)
        return out[:n].copy()

    def depth_image(self, truth, pose, width=640, height=480, max_range=7.0):
        """16UC1 depth frame (mm, 0 = no return) for `pose` = (x, y, z, yaw, pitch)."""
        img = np.zeros((height, width), dtype=np.uint16)
        s = width / 640.0
        self.L.synth_depth_image(C.byref(self.g), truth.ctypes.data, (C.c_double * 5)(*pose), width, height,
                                 C.c_double(CAM["fx"] * s), C.c_double(CAM["fy"] * s), C.c_double(CAM["cx"] * s),
                                 C.c_double(CAM["cy"] * s), C.c_double(max_range), img.ctypes.data)
        return img

    @staticmethod
    def pose_quaternion(pose):
        """Orientation quaternion (w, x, y, z) of the camera frame of `pose`: columns of R are the
        camera's x (right), y (down) and z (forward) axes in the world."""
        yaw, pitch = pose[3], pose[4]
        cyw, syw, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        R = np.array([[syw, cyw * sp, cyw * cp], [-cyw, syw * sp, syw * cp], [0.0, -cp, sp]])
        tr = np.trace(R)
        if tr > 0:
            s4 = np.sqrt(tr + 1.0) * 2
            q = [0.25 * s4, (R[2, 1] - R[1, 2]) / s4, (R[0, 2] - R[2, 0]) / s4, (R[1, 0] - R[0, 1]) / s4]
        else:
            i = int(np.argmax(np.diag(R)))
            j, k = (i + 1) % 3, (i + 2) % 3
            s4 = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
            q = [0.0] * 4
            q[0] = (R[k, j] - R[j, k]) / s4
            q[1 + i] = 0.25 * s4
            q[1 + j] = (R[j, i] + R[i, j]) / s4
            q[1 + k] = (R[k, i] + R[i, k]) / s4
        q = np.array(q)
        return q / np.linalg.norm(q)
