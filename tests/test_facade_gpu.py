"""The C++ facade (fuel_amd/facade: fast_planner::SDFMap / EDTEnvironment / FrontierFinder /
BsplineOptimizer with the reference's signatures) driven the way the reference's callers drive
the originals, checked against the oracle.  Includes the host-mirror contract: the inline getters
compiled into the CALLER read occupancy_buffer_/..._inflate_/distance_buffer_ on the host."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import fuel_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "fuel_amd", "facade", "facade_demo")


def test_facade_cycle_matches_oracle(tmp_path):
    assert os.path.exists(DEMO), "facade_demo not built (run __graft_entry__.build())"
    map_size = (10.0, 8.0, 4.0)
    box = ((-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    cmin = 10
    om = fo.OracleMap(map_size, *box)
    truth = om.fixture_world(3, 14)
    frames = []
    for k in range(8):
        pose = om.fixture_camera(truth, 5, k, 8, 0.6)
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
        frames.append((pts, pose[:3].copy()))
    extra = []
    for k in range(3):
        pose = om.fixture_camera(truth, 24, k, 3, 0.6)  # round 2 drops four clusters (positions 1,7,7,7) and adds new ones
        extra.append((om.fixture_render(truth, pose, 160, 120, 2, 2), pose[:3].copy()))
    rng = np.random.default_rng(2)
    N = 12
    a = np.array([-3.0, -2.0, 1.0])
    b = np.array([3.0, 2.0, 1.2])
    ctrl = a + (b - a) * np.linspace(0, 1, N)[:, None] + rng.normal(scale=0.2, size=(N, 3))
    dt = 0.4
    st = np.zeros((3, 3))
    en = np.zeros((3, 3))
    st[0] = (ctrl[0] + 4 * ctrl[1] + ctrl[2]) / 6
    en[0] = (ctrl[-1] + 4 * ctrl[-2] + ctrl[-3]) / 6
    scen = tmp_path / "scen.bin"
    with open(scen, "wb") as f:
        f.write(struct.pack("10d", *map_size, *box[0], *box[1], float(cmin)))
        f.write(struct.pack("i", len(frames)))
        for pts, cam in frames:
            f.write(struct.pack("i", len(pts)))
            f.write(struct.pack("3d", *cam))
            f.write(np.ascontiguousarray(pts, np.float32).tobytes())
        f.write(struct.pack("i", N))
        f.write(struct.pack("d", dt))
        f.write(ctrl.astype(np.float64).tobytes())
        f.write(st.tobytes())
        f.write(en.tobytes())
        f.write(struct.pack("i", len(extra)))  # second round of the tour-planning part
        for pts, cam in extra:
            f.write(struct.pack("i", len(pts)))
            f.write(struct.pack("3d", *cam))
            f.write(np.ascontiguousarray(pts, np.float32).tobytes())
    res = tmp_path / "res.bin"
    p = subprocess.run([DEMO, str(scen), str(res)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    # oracle side, same call sequence
    for pts, cam in frames:
        om.input_points(pts, cam)
        om.inflate_local()
        om.update_esdf()
    raw = open(res, "rb").read()
    off = 0
    (n,) = struct.unpack_from("i", raw, off)
    off += 4
    assert n == om.N
    occ = np.frombuffer(raw, np.int8, n, off)
    off += n
    infl = np.frombuffer(raw, np.int8, n, off)
    off += n
    dist = np.frombuffer(raw, np.float64, n, off)
    off += 8 * n
    state = np.where(om.occ < om.l_min - 1e-3, 0, np.where(om.occ > om.l_occ, 2, 1))
    assert np.array_equal(occ, state)
    assert np.array_equal(infl, om.infl)
    assert np.abs(np.minimum(dist, 1e6) - np.minimum(om.dist, 1e6)).max() <= 1e-4
    ub = om.get_updated_box(reset=False)
    of = fo.OracleFrontier(om, cmin)
    n_o = of.search()
    (nc,) = struct.unpack_from("i", raw, off)
    off += 4
    assert nc == n_o
    for c in of.clusters(0):
        (sz,) = struct.unpack_from("i", raw, off)
        off += 4
        pos = np.frombuffer(raw, np.float64, 3 * sz, off).reshape(sz, 3)
        off += 24 * sz
        idx = np.floor((pos - om.origin) * 10.0).astype(np.int64)
        adr = (idx[:, 0] * om.nvox[1] + idx[:, 1]) * om.nvox[2] + idx[:, 2]
        assert np.array_equal(np.sort(adr), np.sort(c))
    f0, f1 = struct.unpack_from("2d", raw, off)
    off += 16
    (ng,) = struct.unpack_from("i", raw, off)
    off += 4
    g0 = np.frombuffer(raw, np.float64, ng, off)
    x0 = np.concatenate([ctrl.reshape(-1), [dt]])
    cf = fo.COST["NORMAL_PHASE"] | fo.COST["MINTIME"]
    co, go = fo.bspline_cost_grad(om, x0, N, cf, fo.bspline_pt_dist(ctrl), st, en, 3, 3, dt)
    assert abs(f0 - co) <= 1e-6 * max(1.0, abs(co))
    assert np.abs(g0 - go).max() <= 1e-4
    assert f1 < f0  # the facade's solver loop decreases the reference objective
    off += 8 * ng
    # second finder: split + viewpoints + top-viewpoint query, against the literal oracle (the reference's own BFS
    # order): the facade's default is frontier/reference_order = 2, exact for searches of this size
    na, nd, ntop, covered = struct.unpack_from("4i", raw, off)
    off += 16
    top = np.frombuffer(raw, np.float64, 7 * ntop, off).reshape(ntop, 7)
    of2 = fo.OracleFrontier(om, cmin, cluster_size_xy=1.0, down_sample=3, split=True)
    of2.set_viewpoint_cfg(fo.viewpoint_cfg(min_visib_num=3))
    om.set_updated_box(*ub)
    assert of2.search() > 0
    of2.compute_to_visit()
    assert na == len(of2.clusters(1)) > 0 and nd == len(of2.clusters(2)) and ntop == na
    cur = np.array([0.0, 0.0, 1.0])
    for k in range(na):
        py, vis = of2.viewpoints(1, k)
        pick = py[0]
        for v in py:
            if np.linalg.norm(v[:3] - cur) < 0.75:
                continue
            pick = v
            break
        assert np.array_equal(top[k, :3], pick[:3]) and abs(top[k, 3] - pick[3]) <= 1e-9
        assert np.array_equal(top[k, 4:], of2.cluster_info(1, k)[0])
    assert covered == int(of2.is_covered())
    off += 56 * ntop
    # tour planning: cost matrix kept incrementally over a second round (clusters dropped and added);
    # with the demo's ViewNode (straight flight + 0.1 |yaw difference|, path = the two end points) the
    # matrix must equal that formula between the best viewpoints of the FINAL active list, in list order
    n_extra, n_removed, dim, n_path = struct.unpack_from("4i", raw, off)
    off += 16
    mat = np.frombuffer(raw, np.float64, dim * dim, off).reshape(dim, dim)
    off += 8 * dim * dim
    tpath = np.frombuffer(raw, np.float64, 3 * n_path, off).reshape(n_path, 3)
    for pts, cam in extra:
        om.input_points(pts, cam)
        om.inflate_local()
    of2.search()
    assert n_extra == len(extra) and n_removed == len(of2.removed_ids())
    assert n_removed > 0, "scenario no longer drops a cluster in round 2: the incremental bookkeeping is not exercised"
    of2.compute_to_visit()
    tops = [of2.viewpoints(1, k)[0][0] for k in range(len(of2.clusters(1)))]
    assert dim == len(tops) + 1 and dim > 2

    def cost(p1, y1, p2, y2):
        return np.linalg.norm(p2 - p1) + 0.1 * abs(y2 - y1)

    want = np.zeros((dim, dim))
    for i, a in enumerate(tops):
        want[0, i + 1] = cost(cur, 0.3, a[:3], a[3])
        for j, b in enumerate(tops):
            if i != j:
                want[i + 1, j + 1] = cost(a[:3], a[3], b[:3], b[3])
    assert np.abs(mat - want).max() <= 1e-9
    tour = [(k * 3) % (dim - 1) for k in range(min(4, dim - 1))]
    wpath = [cur, tops[tour[0]][:3]]
    for a, b in zip(tour[:-1], tour[1:]):
        wpath += [tops[a][:3], tops[b][:3]] if a != b else []
    assert n_path == len(wpath) and np.abs(tpath - np.array(wpath)).max() <= 1e-12
