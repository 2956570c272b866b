"""Pins the oracle against the REAL reference: oracle/_ref/libfuel_ref.so is built from
/root/reference/fuel_planner/{plan_env/src/sdf_map.cpp, raycast.cpp, edt_environment.cpp,
bspline_opt/src/bspline_optimizer.cpp} with header stand-ins for Eigen/ROS/PCL/NLopt
(oracle/ref_build/).  Bars: bit-exact (same IEEE f64 operations in the same order) for fusion,
inflation, ESDF, ray walking; <= 1e-12 relative for the B-spline cost/gradient (summation order
of independent terms is the only difference).  Skipped where the library was never built."""
import os
import sys

import numpy as np
import pytest

import helpers
from oracle import fuel_oracle as fo
from oracle.ref_build import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def twin(map_size, box, **kw):
    return fo.OracleMap(map_size, *box, **kw), ref.RefMap(map_size, *box, **kw)


def test_constants_and_initial_state():
    om, rm = twin((50.0, 50.0, 10.0), ((-10, -15, 0), (10, 15, 2)))
    assert om.nvox == rm.nvox == (500, 500, 100)
    assert np.array_equal(om.occ, rm.occ) and np.array_equal(om.dist, rm.dist)


@pytest.mark.parametrize("optimistic,signed", [(0, 0), (1, 0), (1, 1)])
def test_fusion_inflation_esdf_bit_exact(optimistic, signed):
    box = ((-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    om, rm = twin((10.0, 8.0, 4.0), box, optimistic=optimistic, signed_dist=signed)
    truth = om.fixture_world(3, 14)
    rng = np.random.default_rng(9)
    for k in range(24):
        pose = om.fixture_camera(truth, 5, k, 24, 0.9)
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2, maxdist=9.0 if k % 3 == 0 else 5.0)
        extra = pose[:3] + rng.normal(scale=6.0, size=(40, 3))
        pts = np.vstack([pts, extra.astype(np.float32), pts[:50]])
        om.input_points(pts, pose[:3])
        rm.input_points(pts, pose[:3])
        assert om.get_local_bound() == rm.get_local_bound()
        assert np.array_equal(np.concatenate(om.get_updated_box()), np.concatenate(rm.get_updated_box()))
        if k % 6 == 5:
            om.inflate_local()
            rm.inflate_local()
            om.update_esdf()
            rm.update_esdf()
            assert np.array_equal(om.occ, rm.occ)
            assert np.array_equal(om.infl, rm.infl)
            assert np.array_equal(om.dist, rm.dist)  # identical doubles, incl. res*sqrt(DBL_MAX)
    pos = om.origin - 0.3 + (np.array([10.0, 8.0, 4.0]) + 0.6) * rng.random((4000, 3))
    d0, g0 = om.dist_grad(pos)
    d1, g1 = rm.dist_grad(pos)
    assert np.array_equal(d0, d1) and np.array_equal(g0, g1)


def test_char_wrap_of_raycast_num():
    om, rm = twin((4.0, 4.0, 2.0), ((-2, -2, -1), (2, 2, 1)))
    cam = np.array([0.0, 0.0, 0.0])
    rng = np.random.default_rng(1)
    for k in range(300):  # crosses the 127 -> -128 and the == -1 frames
        pts = (rng.random((30, 3)) * np.array([3.6, 3.6, 1.6]) - np.array([1.8, 1.8, 0.8])).astype(np.float32)
        om.input_points(pts, cam)
        rm.input_points(pts, cam)
        if k in (126, 127, 128, 254, 255, 256, 299):
            assert np.array_equal(om.occ, rm.occ), k


def test_full_box_and_map_face_wrap():
    om, rm = twin((4.0, 3.0, 2.0), ((-2, -1.5, -1), (2, 1.5, 1)))
    nv = om.nvox
    for m in (om, rm):
        occ = m.occ.reshape(nv)
        for id3 in [(0, 0, 0), (0, 0, nv[2] - 1), (0, nv[1] - 1, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1),
                    (5, 0, 7), (5, nv[1] - 1, 7), (9, 11, 0), (9, 11, nv[2] - 1), (nv[0] - 1, 3, 3)]:
            occ[id3] = 2.0
        m.set_local_bound(*helpers.full_box(nv))
        m.inflate_local()
        m.update_esdf()
    assert np.array_equal(om.infl, rm.infl) and np.array_equal(om.dist, rm.dist)


@pytest.mark.parametrize("ceil_h", [1.5, 1.55, 2.95])
def test_virtual_ceiling_matches_reference(ceil_h):
    """clearAndInflateLocalMap's virtual ceiling (sdf_map.cpp:464-471; enabled at 3.2 m by kino_algorithm.xml:75 /
    topo_algorithm.xml:70): occupancy_buffer_[x, y, ceil_id] = clamp_max_log over the x,y extent of the local
    bound -- whatever the bound's z range -- AFTER the stamps, so the ceiling row is inflated only by the NEXT
    call.  Fusion keeps updating the ceiling voxels in between (misses pull them down again)."""
    box = ((-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    om, rm = twin((10.0, 8.0, 4.0), box, virtual_ceil_height=ceil_h)
    truth = om.fixture_world(3, 14)
    for k in range(12):
        pose = om.fixture_camera(truth, 5, k, 12, 0.9)
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
        for m in (om, rm):
            m.input_points(pts, pose[:3])
            m.inflate_local()
            if k % 3 == 2:
                m.update_esdf()
        assert np.array_equal(om.occ, rm.occ), k
        assert np.array_equal(om.infl, rm.infl), k
        assert np.array_equal(om.dist, rm.dist), k
    ceil_id = int(np.floor((ceil_h - om.origin[2]) * (1.0 / om.res)))
    row = om.occ.reshape(om.nvox)[:, :, ceil_id]
    assert (row == om.l_max).sum() > 500  # the ceiling is really there
    # bounds whose z range ends AT the ceiling row, below it and above it: the row is written all the same
    for zhi in (ceil_id, ceil_id - 3, om.nvox[2] - 1):
        for m in (om, rm):
            m.occ.reshape(m.nvox)[20:40, 10:30, ceil_id] = om.l_min  # knock a hole, let the next call close it
            m.set_local_bound((15, 5, 2), (45, 35, zhi))
            m.inflate_local()
            m.update_esdf()
        assert np.array_equal(om.occ, rm.occ) and np.array_equal(om.infl, rm.infl) and np.array_equal(om.dist, rm.dist)
        assert np.all(om.occ.reshape(om.nvox)[20:40, 10:30, ceil_id] == om.l_max)


def test_fusion_from_outside_the_map_matches_reference():
    """camera beyond the +z / +y faces, end points inside the map: ray cells past a face alias voxels of the next
    row / slab (setCacheOccupancy has no bounds test at all, sdf_map.cpp:243-257); oracle == reference on the
    part of that situation where every address stays in [0, N) (anything else is undefined behaviour there)."""
    from test_gpu_parity_r2 import outside_camera_frames
    map_size = (6.0, 5.0, 3.0)
    om, rm = twin(map_size, (None, None))
    for k, (pts, cam) in enumerate(outside_camera_frames(om.origin, map_size)):
        om.input_points(pts, cam)
        rm.input_points(pts, cam)
        assert om.get_local_bound() == rm.get_local_bound()
        assert np.array_equal(om.occ, rm.occ), k
    assert (om.occ != om.occ.min()).sum() > 10000


def test_raycaster_cells_identical():
    om, rm = twin((8.0, 6.0, 4.0), ((-3, -2, 0), (3, 2, 2)))
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = om.origin + 0.2 + (np.array([8.0, 6.0, 4.0]) - 0.4) * rng.random(3)
        b = om.origin + 0.2 + (np.array([8.0, 6.0, 4.0]) - 0.4) * rng.random(3)
        assert np.array_equal(om.raycast_cells(a, b), rm.raycast_cells(a, b))


@pytest.mark.parametrize("cf", [0x11F, 0x01F, 0x039, 0x041, 0x1FF, 0x002, 0x104])
def test_bspline_combine_cost_matches_reference(cf):
    om, rm = twin((20.0, 20.0, 5.0), ((-9, -9, 0), (9, 9, 3)))
    truth = om.fixture_world(42, 60)
    om.fixture_known_state(truth, 42, 10)
    rm.occ[:] = om.occ
    for m in (om, rm):
        m.set_local_bound(*helpers.full_box(om.nvox))
        m.inflate_local()
        m.update_esdf()
    rng = np.random.default_rng(4)
    for N in (6, 14, 32):
        ctrl = helpers.make_trajectories(rng, 6, N, np.array([-8.5, -8.5, 0.5]), np.array([8.5, 8.5, 2.5]))
        mint = bool(cf & 0x100)
        x, ptd, st, en = helpers.bspline_inputs(ctrl, 0.175, mint)
        for c in range(len(ctrl)):
            guide = ctrl[c, 3:N - 3] + 0.1 if N > 6 else np.zeros((0, 3))
            widx = np.array([1, N // 2, N - 3], dtype=np.int32)
            kw = dict(guide_pts=guide, waypoints=ctrl[c, widx + 1] + 0.2, waypt_idx=widx,
                      view=(ctrl[c, N // 2] + 0.5, np.array([0.5, 1.0, 0.2]), N // 2 + 1), ld_view=0.7)
            for end_n in (1, 2, 3):
                f0, g0 = fo.bspline_cost_grad(om, x[c], N, cf, ptd[c], st[c], en[c], end_n, 3, 0.175,
                                              1.0 if mint else -1.0, **kw)
                f1, g1 = ref.bspline_cost_grad(rm, x[c], N, cf, ptd[c], st[c], en[c], end_n, 3, 0.175,
                                               1.0 if mint else -1.0, **kw)
                assert abs(f0 - f1) <= 1e-12 * max(1.0, abs(f1))
                assert np.abs(g0 - g1).max() <= 1e-12 * max(1.0, np.abs(g1).max())


def test_frontier_search_matches_reference_incrementally():
    """FrontierFinder::searchFrontiers of the reference (BFS order) vs the oracle: identical cells in
    identical (BFS) order, identical flags, removed ids and cluster info, over several rounds."""
    map_size = (20.0, 20.0, 5.0)
    box = ((-9.0, -9.0, 0.0), (9.0, 9.0, 3.0))
    om, rm = twin(map_size, box)
    truth = om.fixture_world(42, 60)
    of = fo.OracleFrontier(om, 100)
    rf = ref.RefFrontier(rm, 100)
    k = 0
    for r in range(5):
        for _ in range(12):
            pose = om.fixture_camera(truth, 7, k, 60, 0.7)
            k += 1
            pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
            om.input_points(pts, pose[:3])
            rm.input_points(pts, pose[:3])
        assert of.search() == rf.search()
        for a, b in zip(of.clusters(0), rf.clusters(0)):
            assert np.array_equal(a, b)  # same cells in the same BFS order
        assert np.array_equal(of.flags, rf.flags)
        assert np.array_equal(of.removed_ids(), rf.removed_ids())
        for c in range(len(of.clusters(0))):
            for u, v in zip(of.cluster_info(0, c), rf.cluster_info(0, c)):
                assert np.array_equal(u, v)
        of.commit(r == 2)
        rf.commit(r == 2)


def test_frontier_low_z_seeds_box_faces_and_small_clusters_match_reference():
    map_size = (8.0, 6.0, 4.0)
    box = ((-2.0, -1.5, -0.5), (1.0, 2.0, 1.0))
    om, rm = twin(map_size, box)
    truth = om.fixture_world(5, 6)
    om.fixture_known_state(truth, 5, 6, 1.0, 2.2)
    rm.occ[:] = om.occ
    for cmin in (0, 5, 60):
        of = fo.OracleFrontier(om, cmin)
        rf = ref.RefFrontier(rm, cmin)
        for m in (om, rm):
            m.set_updated_box((-1.0, -1.0, 0.2), (0.5, 1.0, 0.8))
        assert of.search() == rf.search()
        for a, b in zip(of.clusters(0), rf.clusters(0)):
            assert np.array_equal(a, b)
        assert np.array_equal(of.flags, rf.flags)


def test_golden_fixture_agrees_with_reference():
    """The committed known-answer vectors were produced by the oracle; the reference build must
    reproduce them from the same stored inputs."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "small_cycle.npz"))
    rm = ref.RefMap(mg.MAP_SIZE, *mg.BOX)
    for k in range(mg.N_FRAMES):
        rm.input_points(z["pts%d" % k], z["cam%d" % k])
    assert np.array_equal(rm.occ, z["occupancy"])
    lo, hi = rm.get_local_bound()
    assert np.array_equal(np.array([lo, hi]), z["local_bound"])
    rm.inflate_local()
    rm.update_esdf()
    assert np.array_equal(rm.infl, z["inflate"])
    sl = tuple(slice(lo[i], hi[i] + 1) for i in range(3))
    assert np.array_equal(rm.dist.reshape(rm.nvox)[sl], z["distance_box"])
    d, g = rm.dist_grad(z["query_pos"])
    assert np.array_equal(d, z["query_dist"]) and np.array_equal(g, z["query_grad"])
    rf = ref.RefFrontier(rm, mg.CLUSTER_MIN)
    n = rf.search()
    off = z["cluster_offsets"]
    assert n == len(off) - 1
    for k, c in enumerate(rf.clusters(0)):
        assert np.array_equal(np.sort(c), z["cluster_cells"][off[k]:off[k + 1]])
    assert np.array_equal(rf.flags, z["frontier_flags"])
    ctrl, st, en = z["ctrl"], z["start"], z["end"]
    for c in range(len(ctrl)):
        x = np.concatenate([ctrl[c].reshape(-1), [0.2]])
        f, gr = ref.bspline_cost_grad(rm, x, ctrl.shape[1], 0x11F, fo.bspline_pt_dist(ctrl[c]), st[c], en[c], 3, 3, 0.2)
        assert abs(f - z["bspline_cost"][c]) <= 1e-12 * abs(f)
        assert np.abs(gr - z["bspline_grad"][c]).max() <= 1e-12 * np.abs(gr).max()


def _depth_frames(seed=3, n=6, width=160, height=120):
    """Synthetic 16UC1 frames with no-return pixels (0), near hits (< mindist) and far hits (> maxdist)."""
    from fuel_amd import synth
    w = synth.World.for_map_size((10.0, 8.0, 4.0))
    truth = w.world(seed, 14)
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        pose = w.camera(truth, 5, k, n, 0.9)
        img = w.depth_image(truth, pose, width, height, max_range=7.0)
        img[rng.random(img.shape) < 0.02] = 0                       # dropouts
        img[rng.random(img.shape) < 0.01] = rng.integers(1, 199)    # closer than depth_filter_mindist
        out.append((img, pose, synth.World.pose_quaternion(pose)))
    return out


@pytest.mark.skipif(not ref.mapros_available(), reason="oracle/_ref/libfuel_ref_mapros.so not built")
@pytest.mark.parametrize("margin,skip", [(2, 2), (2, 1), (3, 3), (1, 1)])
def test_depth_projection_bit_exact_against_real_map_ros(margin, skip):
    """proessDepthImage (map_ros.cpp:176-215): same float points in the same order, including the
    'zero test one sample ahead' quirk.  Configurations keep u+skip inside the row (skip <= margin) or
    at least inside the image, where the reference's read is defined."""
    total = 0
    for img, pose, q in _depth_frames():
        s = img.shape[1] / 640.0
        cfg = fo.depth_cfg(fx=387.229248046875 * s, fy=387.229248046875 * s, cx=321.04638671875 * s,
                           cy=243.44969177246094 * s, margin=margin, skip=skip)
        a = fo.project_depth(img, pose[:3], q, cfg)
        b = ref.project_depth(img, pose[:3], q, cfg)
        assert a.shape == b.shape and np.array_equal(a, b)
        total += len(a)
    assert total > 8000


@pytest.mark.skipif(not ref.mapros_available(), reason="oracle/_ref/libfuel_ref_mapros.so not built")
def test_depth_projection_quirk_is_real():
    """A zero one sample AHEAD of a valid pixel turns that pixel into a max-range miss."""
    img = np.full((12, 16), 1500, dtype=np.uint16)
    img[4, 8] = 0
    cfg = fo.depth_cfg(fx=20.0, fy=20.0, cx=8.0, cy=6.0, margin=2, skip=2)
    q = (1.0, 0.0, 0.0, 0.0)
    a = fo.project_depth(img, (0, 0, 0), q, cfg)
    b = ref.project_depth(img, (0, 0, 0), q, cfg)
    assert np.array_equal(a, b)
    # rows v = 2,4,6,8 x columns u = 2,...,12; in row v=4: u=6 reads the zero at u=8 -> max range;
    # u=8 itself has depth 0 but passes the zero test (reads u=10) and is dropped by the min filter
    assert len(a) == 23
    assert np.array_equal(a[6:11, 2], np.float32([1.5, 1.5, 5.0, 1.5, 1.5]))


def _explored_pair(seed=42, n_frames=28):
    """Same explored state in the oracle map and the real SDFMap (own fusion each; bit-equal)."""
    om, truth, frames, box = helpers.explored_oracle_map((16.0, 14.0, 4.0), 30, n_frames, seed=seed)
    rm = ref.RefMap((16.0, 14.0, 4.0), *box)
    for pts, cam in frames:
        rm.input_points(pts, cam)
    assert np.array_equal(om.occ, rm.occ)
    return om, rm


@pytest.mark.parametrize("seed,size_xy", [(42, 2.0), (7, 1.2), (11, 3.0)])
def test_split_large_frontiers_against_real_reference(seed, size_xy):
    """searchFrontiers INCLUDING splitLargeFrontiers (frontier_finder.cpp:166-242,374-390,757-774) run by
    the real frontier_finder.cpp (with the VoxelGrid / EigenSolver stand-ins of compat/) equals the oracle:
    same clusters in the same order, same cell order (BFS order survives the partition), bit-equal
    averages / boxes / filtered cells."""
    om, rm = _explored_pair(seed)
    of = fo.OracleFrontier(om, cluster_min=60, cluster_size_xy=size_xy, down_sample=3, split=True)
    rf = ref.RefFrontier(rm, cluster_min=60, cluster_size_xy=size_xy)
    plain = fo.OracleFrontier(om, cluster_min=60)
    ub = om.get_updated_box(reset=False)
    for m_ in (om, rm):
        m_.set_updated_box(*ub)
    n0 = plain.search()
    om.set_updated_box(*ub)
    n1, n2 = of.search(), rf.search()
    assert n1 == n2 and n1 > n0 > 0  # something was actually split
    ca, cb = of.clusters(0), rf.clusters(0)
    for k in range(n1):
        assert np.array_equal(ca[k], cb[k])
        for x, y in zip(of.cluster_info(0, k), rf.cluster_info(0, k)):
            assert np.array_equal(x, y)
        fa_, fb_ = of.filtered(0, k), rf.filtered(0, k)
        assert fa_.shape == fb_.shape and np.array_equal(fa_, fb_) and len(fa_) > 0
    # the split is a partition of the region-grown clusters
    assert np.array_equal(np.sort(np.concatenate(ca)), np.sort(np.concatenate(plain.clusters(0))))


@pytest.mark.parametrize("seed,size_xy", [(42, 2.0), (7, 1.2)])
def test_viewpoint_sampling_against_real_reference(seed, size_xy):
    """computeFrontiersToVisit / sampleViewpoints / countVisibleCells / isNearUnknown (frontier_finder.cpp:
    392-423,662-755) with the REAL perception_utils.cpp: same frontiers_/dormant split, and per cluster the
    same viewpoints (bit-equal positions, yaws, coverage counts, same order after the coverage sort);
    then isFrontierCovered (:697-719) after further fusion."""
    om, rm = _explored_pair(seed)
    for m_ in (om, rm):
        m_.inflate_local()
    assert np.array_equal(om.infl, rm.infl)
    vcfg = fo.viewpoint_cfg(min_visib_num=5)
    of = fo.OracleFrontier(om, cluster_min=60, cluster_size_xy=size_xy, down_sample=3, split=True)
    of.set_viewpoint_cfg(vcfg)
    rf = ref.RefFrontier(rm, cluster_min=60, cluster_size_xy=size_xy, viewpoint_cfg=vcfg)
    ub = om.get_updated_box(reset=False)
    rm.set_updated_box(*ub)
    assert of.search() == rf.search() > 0
    of.compute_to_visit()
    rf.compute_to_visit()
    n_act, n_dor = len(of.clusters(1)), len(of.clusters(2))
    assert n_act == len(rf.clusters(1)) and n_dor == len(rf.clusters(2)) and n_act > 0
    total = 0
    for k in range(n_act):
        (pa, va), (pb, vb) = of.viewpoints(1, k), rf.viewpoints(1, k)
        assert np.array_equal(va, vb) and np.array_equal(pa, pb) and len(va) > 0
        assert np.all(va[:-1] >= va[1:])
        total += len(va)
    assert total > 20
    for a, b in zip(of.clusters(1) + of.clusters(2), rf.clusters(1) + rf.clusters(2)):
        assert np.array_equal(a, b)
    # isFrontierCovered: nothing changed yet -> False; after more fusion near a frontier -> same answer
    assert of.is_covered() == rf.is_covered()
    truth = om.fixture_world(seed, 30)
    flips = 0
    for k in range(6):
        pose = om.fixture_camera(truth, 99, k, 6, 0.6)
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
        om.input_points(pts, pose[:3])
        rm.input_points(pts, pose[:3])
        a, b = of.is_covered(), rf.is_covered()
        assert a == b
        flips += int(a)
    assert flips > 0


# ---- NonUniformBspline glue (bspline/src/non_uniform_bspline.cpp compiled unmodified) ----
def _spline_case(seed, K, ts):
    rng = np.random.default_rng(seed)
    pts = np.cumsum(rng.normal(scale=0.3, size=(K, 3)), axis=0) + np.array([1.0, -2.0, 1.0])
    der = rng.normal(scale=0.8, size=(4, 3))
    return pts, der


@pytest.mark.parametrize("degree", [3, 4, 5])
@pytest.mark.parametrize("K", [2, 3, 9, 40])
def test_spline_parameterize_matches_reference(degree, K):
    ts = 0.17 + 0.05 * degree
    pts, der = _spline_case(100 * degree + K, K, ts)
    want = ref.spline_parameterize(ts, pts, der, degree)
    got = fo.spline_parameterize(ts, pts, der, degree)
    assert got.shape == (K + degree - 1, 3)
    # the reference's own solver is Eigen's QR (stood in by a Givens QR in the build); the oracle restates
    # column-pivoted Householder: two different orthogonal factorizations of a system with cond < 1e3
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-11)


@pytest.mark.parametrize("degree,ks,ke", [(3, 2, 0), (3, 2, 2), (4, 3, 1), (5, 2, 2), (3, 0, 0)])
def test_spline_boundary_states_match_reference(degree, ks, ke):
    ts = 0.23
    pts, der = _spline_case(7 + degree, 14, ts)
    ctrl = fo.spline_parameterize(ts, pts, der, degree)
    s0, e0 = ref.spline_boundary_states(ctrl, ts, degree, ks, ke)
    s1, e1 = fo.spline_boundary_states(ctrl, ts, degree, ks, ke)
    np.testing.assert_array_equal(s1, s0)  # same arithmetic in the same order: bit-exact
    np.testing.assert_array_equal(e1, e0)
    if degree == 5:  # exactly determined system up to rounding: the spline interpolates its constraints
        np.testing.assert_allclose(s1[0], pts[0], atol=1e-9)
        np.testing.assert_allclose(s1[1], der[0], atol=1e-9)
        np.testing.assert_allclose(e1[0], pts[-1], atol=1e-9)


# ---- tour planning bookkeeping of the REAL FrontierFinder (updateFrontierCostMatrix & co.) ----
def test_reference_cost_matrix_bookkeeping_equals_pairwise_costs():
    """What the facade's updateFrontierCostMatrix / getFullCostMatrix / getPathForTour must reproduce
    (tests/test_facade_gpu.py checks the facade against the same statement): with a deterministic
    ViewNode (ref_frontier_stubs.cpp: straight flight + 0.1 |yaw difference|, path = the end points) the
    incrementally kept matrix of the real class -- across a round that drops clusters (removed_ids_) and
    adds new ones -- equals the pairwise costs between the best viewpoints of the final frontiers_."""
    map_size = (10.0, 8.0, 4.0)
    box = ((-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    rm = ref.RefMap(map_size, *box)
    om = fo.OracleMap(map_size, *box)
    truth = om.fixture_world(3, 14)
    for k in range(8):
        pose = om.fixture_camera(truth, 5, k, 8, 0.6)
        rm.input_points(om.fixture_render(truth, pose, 160, 120, 2, 2), pose[:3])
        rm.inflate_local()
    rf = ref.RefFrontier(rm, 10, 1.0, fo.viewpoint_cfg(min_visib_num=3))
    rf.search()
    rf.compute_to_visit()
    rf.update_cost_matrix()
    n1 = len(rf.clusters(1))
    for k in range(3):
        pose = om.fixture_camera(truth, 24, k, 3, 0.6)
        rm.input_points(om.fixture_render(truth, pose, 160, 120, 2, 2), pose[:3])
        rm.inflate_local()
    rf.search()
    removed = list(rf.removed_ids())
    assert len(removed) >= 2 and n1 > len(removed)
    rf.compute_to_visit()
    rf.update_cost_matrix()
    cur = np.array([0.0, 0.0, 1.0])
    mat = rf.full_cost_matrix(cur, (0, 0, 0), (0.3, 0, 0))
    tops = [rf.viewpoints(1, k)[0][0] for k in range(len(rf.clusters(1)))]
    assert mat.shape == (len(tops) + 1, len(tops) + 1)

    def cost(p1, y1, p2, y2):
        return np.linalg.norm(p2 - p1) + 0.1 * abs(y2 - y1)

    want = np.zeros_like(mat)
    for i, a in enumerate(tops):
        want[0, i + 1] = cost(cur, 0.3, a[:3], a[3])
        for j, b in enumerate(tops):
            if i != j:
                want[i + 1, j + 1] = cost(a[:3], a[3], b[:3], b[3])
    assert np.abs(mat - want).max() <= 1e-12
    tour = [0, 3, 1, 5]
    path = rf.path_for_tour(cur, tour)
    wpath = [cur, tops[0][:3]]
    for a, b in zip(tour[:-1], tour[1:]):
        wpath += [tops[a][:3], tops[b][:3]]
    assert np.array_equal(path, np.array(wpath))


def test_reference_map_ros_compiles_against_the_facade_header():
    """The header-level contract of the drop-in: MapROS is a friend of SDFMap and reads / writes
    md_ and mp_ fields by name (map_ros.cpp:142-170, 250-330).  The reference's own map_ros.cpp, unmodified, must
    compile against fuel_amd/facade/plan_env/sdf_map.h (syntax check; ROS / OpenCV / PCL / Eigen are the same
    header stand-ins the reference build uses)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "/root/reference/fuel_planner/plan_env/src/map_ros.cpp"
    if not os.path.exists(src):
        pytest.skip("reference checkout not present")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-w",
           "-I", os.path.join(root, "fuel_amd", "facade"), "-I", os.path.join(root, "include"),
           "-I", os.path.join(root, "oracle", "ref_build", "shim_ros"), "-I", os.path.join(root, "compat"),
           "-I", "/root/reference/fuel_planner/plan_env/include", src]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]


def _ceiling_world(m, x0, x1, y0, y1, extra_row=0):
    """known free space under a patch of unknown ceiling (world z >= 0.5 m) -- the fixtures of
    tests/test_gpu_parity_r3.py::test_reference_order_at_the_lds_limit_exactly / ..._of_a_sheet_..."""
    nv = m.nvox
    occ = np.full(m.N, m.l_min).reshape(nv)
    occ[x0:x1, y0:y1, 15:] = m.l_min - 0.01
    if extra_row:
        occ[x1, y0:y0 + extra_row, 15:] = m.l_min - 0.01
    m.occ[:] = occ.reshape(-1)


@pytest.mark.parametrize("name", ["lds_limit", "lds_limit_plus_one", "sheet"])
def test_large_cluster_fixtures_of_the_gpu_suite_match_the_real_reference(name):
    """The GPU suite checks the device's cell order of its largest clusters (26 624 / 26 625 cells at the hand-over
    between the two level sweeps, a 608 400-cell sheet) against the oracle; this is the other half of the chain: the
    REAL FrontierFinder::searchFrontiers / expandFrontier (frontier_finder.cpp:54-164) on the same worlds yields the
    oracle's cells in the oracle's order, the same average_ and boxes, the same flags."""
    if name == "sheet":
        map_size, org = (80.0, 80.0, 3.0), (-40.0, -40.0, -1.0)
    else:
        map_size, org = (30.0, 30.0, 3.0), (-15.0, -15.0, -1.0)
    box = ((org[0] + 1.0, org[1] + 1.0, 0.0), (-org[0] - 1.0, -org[1] - 1.0, 1.4))
    om, rm = twin(map_size, box)
    if name == "sheet":
        _ceiling_world(om, 0, om.nvox[0], 0, om.nvox[1])
        cells = 608400
    else:
        _ceiling_world(om, 40, 180, 40, 192, 79 if name == "lds_limit" else 80)
        cells = 26624 if name == "lds_limit" else 26625
    rm.occ[:] = om.occ
    of, rf = fo.OracleFrontier(om, 100), ref.RefFrontier(rm, 100)
    for m in (om, rm):
        m.set_updated_box(*box)
    assert of.search() == rf.search() == 1
    a, b = of.clusters(0)[0], rf.clusters(0)[0]
    assert len(a) == cells
    assert np.array_equal(a, b)
    for u, v in zip(of.cluster_info(0, 0), rf.cluster_info(0, 0)):
        assert np.array_equal(u, v)
    assert np.array_equal(of.flags, rf.flags)
