"""Generates tests/golden/pillar_plumbing.npz -- BASELINE config #1 ("plumbing"): the reference's own
data fixture (uav_simulator/map_generator/resource/pillar.pcd + the ground grid map_publisher.cpp:38-52
adds) seen through a few depth frames, pushed through THE REAL REFERENCE CODE (oracle/_ref: sdf_map.cpp,
raycast.cpp, map_ros.cpp, frontier_finder.cpp, perception_utils.cpp, bspline_optimizer.cpp compiled
with header stand-ins): depth projection -> fusion -> inflation -> ESDF -> frontier search with
splitting -> viewpoints -> one B-spline cost/gradient.

Needs /root/reference (run in the authoring container); the fixture it writes is what travels:

    python tests/golden/make_pillar_golden.py

Stored: the depth images + poses (inputs) and the reference's answers -- SHA-256 of the bit-exact arrays
(log-odds, inflation), a strided sample of the ESDF, frontier clusters as sorted voxel addresses,
viewpoint counts, B-spline cost/gradient.  tests/test_golden_pillar*.py replay it through the oracle
(CPU) and through libfuelmi (GPU).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import fuel_oracle as fo  # noqa: E402

PCD = "/root/reference/uav_simulator/map_generator/resource/pillar.pcd"
MAP_SIZE = (16.0, 30.0, 5.0)
BOX = ((-7.5, -14.5, 0.0), (7.5, 14.5, 2.5))
CLUSTER_MIN = 100
CLUSTER_XY = 2.0
POSES = [(-5.0, -11.0, 1.0, 0.6, 0.0), (-3.0, -8.0, 1.2, 1.2, -0.1), (-1.0, -5.0, 1.0, 0.3, 0.05),
         (1.5, -2.0, 0.9, 1.9, 0.0), (3.0, 2.0, 1.1, 2.8, -0.05), (0.0, 6.0, 1.0, 4.0, 0.0)]
IMG_W, IMG_H = 320, 240


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def world_truth(w):
    """Voxelise the .pcd points + the ground grid of map_publisher.cpp:47-52 on the map grid."""
    pts = np.loadtxt(PCD, skiprows=11, dtype=np.float32).astype(np.float64)
    mmin = np.minimum(pts[:, :2].min(0), 0.0)
    mmax = np.maximum(pts[:, :2].max(0), 0.0)
    gx = np.arange(mmin[0], mmax[0] + 1e-9, 0.1)
    gy = np.arange(mmin[1], mmax[1] + 1e-9, 0.1)
    ground = np.stack(np.meshgrid(gx, gy, indexing="ij"), -1).reshape(-1, 2)
    allp = np.vstack([pts, np.hstack([ground, np.zeros((len(ground), 1))])])
    nv = np.array(w.g.nv[:])
    org = np.array(w.g.origin[:])
    idx = np.floor((allp - org) / w.g.res).astype(np.int64)
    ok = ((idx >= 0) & (idx < nv)).all(1)
    idx = idx[ok]
    truth = np.zeros(nv, dtype=np.uint8)
    truth[idx[:, 0], idx[:, 1], idx[:, 2]] = 1
    return np.ascontiguousarray(truth.reshape(-1))


def inputs():
    from fuel_amd import synth
    w = synth.World.for_map_size(MAP_SIZE)
    truth = world_truth(w)
    frames = []
    for pose in POSES:
        pose = np.array(pose, dtype=np.float64)
        img = w.depth_image(truth, pose, IMG_W, IMG_H, max_range=7.0)
        frames.append((img, pose[:3].copy(), synth.World.pose_quaternion(pose)))
    rng = np.random.default_rng(31)
    a = np.array([-4.0, -9.0, 1.0])
    b = np.array([2.0, 1.0, 1.2])
    ctrl = a + (b - a) * np.linspace(0, 1, 16)[:, None] + rng.normal(scale=0.15, size=(16, 3))
    return frames, ctrl


def depth_cfg(mod):
    s = IMG_W / 640.0
    return mod(fx=387.229248046875 * s, fy=387.229248046875 * s, cx=321.04638671875 * s, cy=243.44969177246094 * s)


def run(backend, frames, ctrl):
    """backend: 'ref' (real reference) or 'oracle'.  Returns the dict of answers."""
    if backend == "ref":
        from oracle.ref_build import ref
        m = ref.RefMap(MAP_SIZE, *BOX)
        project = lambda img, p, q: ref.project_depth(img, p, q, depth_cfg(fo.depth_cfg))  # noqa: E731
        ff = ref.RefFrontier(m, CLUSTER_MIN, CLUSTER_XY, fo.viewpoint_cfg())
        cost_grad = lambda *a, **k: ref.bspline_cost_grad(m, *a, **k)  # noqa: E731
    else:
        m = fo.OracleMap(MAP_SIZE, *BOX)
        project = lambda img, p, q: fo.project_depth(img, p, q, depth_cfg(fo.depth_cfg))  # noqa: E731
        ff = fo.OracleFrontier(m, CLUSTER_MIN, cluster_size_xy=CLUSTER_XY, down_sample=3, split=True)
        ff.set_viewpoint_cfg(fo.viewpoint_cfg())
        cost_grad = lambda *a, **k: fo.bspline_cost_grad(m, *a, **k)  # noqa: E731
    npts = []
    for img, pos, q in frames:
        pts = project(img, pos, q)
        npts.append(len(pts))
        m.input_points(pts, pos)
        m.inflate_local()
    ub = m.get_updated_box(reset=False)
    nv = m.nvox
    m.set_local_bound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
    m.inflate_local()
    m.update_esdf()
    out = {"points_per_frame": np.array(npts, np.int32), "updated_box": np.concatenate(ub),
           "occupancy_sha256": np.array(sha(m.occ)), "inflate_sha256": np.array(sha(m.infl)),
           "known_voxels": np.array(int((m.occ > m.occ.min() + 1e-9).sum())),
           "distance_sample": np.minimum(m.dist[::97], 1e6).astype(np.float64)}
    n = ff.search()
    cl = ff.clusters(0)
    out["cluster_offsets"] = np.cumsum([0] + [len(c) for c in cl]).astype(np.int32)
    out["cluster_cells"] = np.concatenate([np.sort(c) for c in cl]).astype(np.int32)
    ff.compute_to_visit()
    out["n_active"] = np.array(len(ff.clusters(1)))
    out["n_dormant"] = np.array(len(ff.clusters(2)))
    out["viewpoint_counts"] = np.array([len(ff.viewpoints(1, k)[1]) for k in range(len(ff.clusters(1)))], np.int32)
    out["best_visib"] = np.array([ff.viewpoints(1, k)[1][0] for k in range(len(ff.clusters(1)))], np.int32)
    out["best_viewpoint"] = np.array([ff.viewpoints(1, k)[0][0] for k in range(len(ff.clusters(1)))])
    # the complete answer of the reference for the active clusters, in ITS order: cells (BFS order), average_,
    # filtered_cells_, every viewpoint (position, yaw, visib_num_) -- what fuelmi_frontier_cfg.reference_order
    # reproduces bit for bit
    act = ff.clusters(1)
    out["active_offsets"] = np.cumsum([0] + [len(c) for c in act]).astype(np.int32)
    out["active_cells_bfs"] = np.concatenate(act).astype(np.int32)
    out["active_average"] = np.array([ff.cluster_info(1, k)[0] for k in range(len(act))])
    filt = [ff.filtered(1, k).astype(np.float32) for k in range(len(act))]
    out["filtered_offsets"] = np.cumsum([0] + [len(c) for c in filt]).astype(np.int32)
    out["filtered_cells"] = np.concatenate(filt)
    vps = [ff.viewpoints(1, k) for k in range(len(act))]
    out["viewpoint_pos_yaw"] = np.concatenate([v[0] for v in vps])
    out["viewpoint_visib"] = np.concatenate([v[1] for v in vps]).astype(np.int32)
    dt = 0.25
    st = np.zeros((3, 3))
    en = np.zeros((3, 3))
    st[0] = (ctrl[0] + 4 * ctrl[1] + ctrl[2]) / 6
    en[0] = (ctrl[-1] + 4 * ctrl[-2] + ctrl[-3]) / 6
    x = np.concatenate([ctrl.reshape(-1), [dt]])
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    f, g = cost_grad(x, len(ctrl), cf, fo.bspline_pt_dist(ctrl), st, en, 3, 3, dt)
    out["bspline_cost"], out["bspline_grad"] = np.array(f), g
    out["n_clusters"] = np.array(n)
    return out


if __name__ == "__main__":
    frames, ctrl = inputs()
    ref_out = run("ref", frames, ctrl)
    ora_out = run("oracle", frames, ctrl)
    for k in ref_out:  # the oracle must already agree with the reference before the fixture is written
        a, b = ref_out[k], ora_out[k]
        if a.dtype.kind in "US":
            assert str(a) == str(b), k
        elif k in ("bspline_cost", "bspline_grad"):
            assert np.allclose(a, b, rtol=1e-12, atol=1e-12), k
        else:
            assert np.array_equal(a, b), k
    res = dict(ref_out)
    for i, (img, pos, q) in enumerate(frames):
        res["depth%d" % i], res["pos%d" % i], res["quat%d" % i] = img, pos, q
    res["ctrl"] = ctrl
    path = os.path.join(HERE, "pillar_plumbing.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes;", int(res["n_clusters"]), "clusters ->", int(res["n_active"]),
          "active /", int(res["n_dormant"]), "dormant;", int(res["known_voxels"]), "known voxels; points",
          res["points_per_frame"].tolist())
