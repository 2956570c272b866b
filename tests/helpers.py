"""Shared fixtures for the parity tests: seeded synthetic maps built with the ORACLE
(test infrastructure), and helpers to push the same state into the GPU map."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import fuel_oracle as fo  # noqa: E402


def explored_oracle_map(map_size, n_obstacles, n_frames, seed=42, cam_seed=7, width=160, height=120,
                        box_margin=1.0, extent=0.7, **kw):
    """Oracle map whose known region was carved by its own inputPointCloud on synthetic frames.

    Exploration box = map shrunk by box_margin in x,y and z in [0, 0.8*sz-1]."""
    org = (-map_size[0] / 2.0, -map_size[1] / 2.0, -1.0)
    box_min = (org[0] + box_margin, org[1] + box_margin, 0.0)
    box_max = (-org[0] - box_margin, -org[1] - box_margin, max(0.8 * map_size[2] - 1.0, 1.0))
    m = fo.OracleMap(map_size, box_min, box_max, **kw)
    truth = m.fixture_world(seed, n_obstacles)
    frames = []
    for k in range(n_frames):
        pose = m.fixture_camera(truth, cam_seed, k, n_frames, extent)
        pts = m.fixture_render(truth, pose, width, height, 2, 2)
        frames.append((pts, pose[:3].copy()))
        m.input_points(pts, pose[:3])
    return m, truth, frames, (box_min, box_max)


def full_box(nvox):
    return (0, 0, 0), (nvox[0] - 1, nvox[1] - 1, nvox[2] - 1)


def make_trajectories(rng, n_traj, n_pts, lo, hi, seg_len=6.0, noise=0.3):
    """Candidate control-point sets: straight segments of seg_len between seeded points plus
    lateral perturbation; returns ctrl [C][N][3]."""
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


def bspline_inputs(ctrl, dt, mintime=True):
    """NLopt-layout variable vectors + boundary states for a batch of control-point sets."""
    C, N, dim = ctrl.shape
    x = ctrl.reshape(C, N * dim)
    if mintime:
        x = np.concatenate([x, np.full((C, 1), dt)], axis=1)
    pt_dist = np.array([fo.bspline_pt_dist(ctrl[c]) for c in range(C)])
    start = np.zeros((C, 3, 3))
    end = np.zeros((C, 3, 3))
    for c in range(C):
        q = ctrl[c]
        start[c, 0] = (q[0] + 4 * q[1] + q[2]) / 6.0 + 0.05
        start[c, 1] = (q[2] - q[0]) / (2 * dt) * 0.9
        start[c, 2] = (q[0] - 2 * q[1] + q[2]) / (dt * dt) * 0.5
        end[c, 0] = (q[-1] + 4 * q[-2] + q[-3]) / 6.0 - 0.03
        end[c, 1] = 0.1
        end[c, 2] = 0.0
    return np.ascontiguousarray(x), pt_dist, start, end
