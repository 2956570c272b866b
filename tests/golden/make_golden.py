"""Generates tests/golden/small_cycle.npz: known-answer vectors of one small seeded plan cycle
(6 fused depth frames -> inflate -> ESDF -> frontier search -> 4 B-spline cost/gradient
evaluations).  Inputs are stored too, so the GPU tests can replay exactly these bytes.

Produced with the CPU oracle; oracle/ref_build/check_ref.py verifies the same quantities against
the REAL reference sources compiled with header shims (oracle/_ref), which pins the oracle.

    python tests/golden/make_golden.py        # rewrites the fixture
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import fuel_oracle as fo  # noqa: E402

MAP_SIZE = (6.0, 5.0, 3.0)
BOX = ((-2.5, -2.0, 0.0), (2.5, 2.0, 1.6))
CLUSTER_MIN = 15
N_FRAMES = 6


def inputs():
    m = fo.OracleMap(MAP_SIZE, *BOX)
    truth = m.fixture_world(11, 8)
    frames = []
    for k in range(N_FRAMES):
        pose = m.fixture_camera(truth, 13, k, N_FRAMES, 0.55)
        pts = m.fixture_render(truth, pose, 128, 96, 2, 2)
        frames.append((pts, pose[:3].copy()))
    rng = np.random.default_rng(17)
    ctrl = np.array(BOX[0]) + 0.4 + (np.array(BOX[1]) - np.array(BOX[0]) - 0.8) * rng.random((4, 10, 3))
    st = rng.normal(size=(4, 3, 3)) * 0.5
    en = rng.normal(size=(4, 3, 3)) * 0.5
    return frames, ctrl, st, en


def compute(frames=None, ctrl=None, st=None, en=None):
    if frames is None:
        frames, ctrl, st, en = inputs()
    m = fo.OracleMap(MAP_SIZE, *BOX)
    for pts, cam in frames:
        m.input_points(pts, cam)
    out = {}
    for k, (pts, cam) in enumerate(frames):
        out["pts%d" % k] = pts
        out["cam%d" % k] = cam
    out["ctrl"], out["start"], out["end"] = ctrl, st, en
    out["occupancy"] = m.occ.copy()
    lo, hi = m.get_local_bound()
    out["local_bound"] = np.array([lo, hi], dtype=np.int32)
    out["updated_box"] = np.concatenate(m.get_updated_box())
    m.inflate_local()
    m.update_esdf()
    out["inflate"] = m.infl.copy()
    sl = tuple(slice(lo[i], hi[i] + 1) for i in range(3))
    out["distance_box"] = m.dist.reshape(m.nvox)[sl].copy()
    of = fo.OracleFrontier(m, CLUSTER_MIN)
    of.search()
    cl = [np.sort(c) for c in of.clusters(0)]
    out["cluster_offsets"] = np.cumsum([0] + [len(c) for c in cl]).astype(np.int32)
    out["cluster_cells"] = np.concatenate(cl).astype(np.int32) if cl else np.zeros(0, np.int32)
    out["frontier_flags"] = of.flags.copy()
    dt = 0.2
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    costs, grads = [], []
    for c in range(len(ctrl)):
        x = np.concatenate([ctrl[c].reshape(-1), [dt]])
        f, g = fo.bspline_cost_grad(m, x, ctrl.shape[1], cf, fo.bspline_pt_dist(ctrl[c]), st[c], en[c], 3, 3, dt)
        costs.append(f)
        grads.append(g)
    out["bspline_cost"] = np.array(costs)
    out["bspline_grad"] = np.array(grads)
    rng = np.random.default_rng(23)
    q = np.array(BOX[0]) - 0.3 + (np.array(BOX[1]) - np.array(BOX[0]) + 0.6) * rng.random((64, 3))
    d, g = m.dist_grad(q)
    out["query_pos"], out["query_dist"], out["query_grad"] = q, d, g
    return out


if __name__ == "__main__":
    res = compute()
    path = os.path.join(HERE, "small_cycle.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes;", len(res["cluster_offsets"]) - 1, "clusters,",
          int((res["occupancy"] > -1.99).sum()), "known voxels")
