"""Parity tests proper: the HIP path, called through the C-ABI (fuel_amd.host is a thin ctypes
mirror of the reference classes), against the CPU oracle on the same seeded inputs, against the
committed golden fixture, and -- at BASELINE.json's full 400x400x100 size -- against the oracle
plus size-independent properties.  Bars: bit-exact for occupancy log-odds, inflation, occupancy
state and frontier voxel indices; <= 1e-4 for ESDF values and B-spline gradients; cost rel 1e-6."""
import os

import numpy as np
import pytest

import helpers
from oracle import fuel_oracle as fo

pytestmark = pytest.mark.gpu

ESDF_TOL = 1e-4
GRAD_TOL = 1e-4
BIG = 1e6  # "no source" sentinel clamp (reference: res*sqrt(DBL_MAX); device: +inf)


@pytest.fixture(scope="module")
def fa():
    import fuel_amd
    assert fuel_amd.lib().fuelmi_device_count() > 0, "no GPU visible: the HIP path cannot run"
    return fuel_amd


def gpu_twin(fa, om, box, **kw):
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1], **kw)
    gm.uploadOccupancy(om.occ)
    return gm


def assert_map_equal(om, gm, box_idx=None, esdf=True):
    h = gm.syncHost(occupancy=True, inflate=True, distance=esdf)
    assert np.array_equal(h["occupancy"], om.occ), "occupancy log-odds not bit-exact"
    assert np.array_equal(h["inflate"], om.infl), "inflated occupancy not bit-exact"
    if esdf:
        d_o = np.clip(om.dist, -BIG, BIG).reshape(om.nvox)
        d_g = np.clip(h["distance"], -BIG, BIG).reshape(om.nvox)
        if box_idx is not None:
            sl = tuple(slice(box_idx[0][i], box_idx[1][i] + 1) for i in range(3))
            d_o, d_g = d_o[sl], d_g[sl]
        assert np.abs(d_o - d_g).max() <= ESDF_TOL
    return h


def sorted_clusters(cl):
    return [np.sort(c) for c in cl]


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("optimistic,signed", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_inflate_and_esdf_full_and_sub_box(fa, optimistic, signed):
    om, _, _, box = helpers.explored_oracle_map((8.0, 6.0, 4.0), 12, 20, optimistic=optimistic,
                                                signed_dist=signed)
    gm = gpu_twin(fa, om, box, optimistic=optimistic, signed_dist=signed)
    for lo, hi in [helpers.full_box(om.nvox), ((10, 5, 3), (60, 40, 30)), ((0, 0, 0), (79, 0, 39)),
                   ((33, 17, 9), (33, 17, 9)), ((70, 50, 30), (79, 59, 39))]:
        om.set_local_bound(lo, hi)
        gm.setLocalBound(lo, hi)
        om.inflate_local()
        om.update_esdf()
        gm.clearAndInflateLocalMap()
        gm.updateESDF3d()
        assert_map_equal(om, gm, (lo, hi))
    gm.close()


def test_inflate_wrap_quirk_at_map_faces(fa):
    """Occupied voxels ON the map faces: stamps wrap across rows exactly like the reference's
    linear-address bounds check (sdf_map.cpp:453-458)."""
    om = fo.OracleMap((4.0, 3.0, 2.0))
    nv = om.nvox
    occ = om.occ.reshape(nv)
    for id3 in [(0, 0, 0), (0, 0, nv[2] - 1), (0, nv[1] - 1, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1),
                (5, 0, 7), (5, nv[1] - 1, 7), (9, 11, 0), (9, 11, nv[2] - 1), (nv[0] - 1, 3, 3), (0, 20, 10)]:
        occ[id3] = om.l_max
    gm = gpu_twin(fa, om, ((-2, -1.5, -1), (2, 1.5, 1)))
    for lo, hi in [helpers.full_box(nv), ((0, 0, 0), (6, nv[1] - 1, nv[2] - 1))]:
        om.infl[:] = 0
        gm.resetBuffer()
        om.set_local_bound(lo, hi)
        gm.setLocalBound(lo, hi)
        om.inflate_local()
        gm.clearAndInflateLocalMap()
        om.update_esdf()
        gm.updateESDF3d()
        assert_map_equal(om, gm, (lo, hi))
    gm.close()


def test_esdf_no_sources_and_all_sources(fa):
    om = fo.OracleMap((3.0, 3.0, 2.0), optimistic=1)
    gm = fa.SDFMap((3.0, 3.0, 2.0), optimistic=1)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.update_esdf()
    gm.updateESDF3d()
    d = gm.syncHost(distance=True)["distance"]
    assert np.all(d == 0.1 * np.sqrt(np.finfo(np.float64).max)) and np.array_equal(d, om.dist)
    gm.close()
    om = fo.OracleMap((3.0, 3.0, 2.0), optimistic=0)  # every voxel unknown -> every voxel a source
    gm = fa.SDFMap((3.0, 3.0, 2.0), optimistic=0)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.update_esdf()
    gm.updateESDF3d()
    assert np.all(gm.syncHost(distance=True)["distance"] == 0.0) and np.all(om.dist == 0.0)
    gm.close()


def test_reset_set_occupied_update_recipe(fa):
    """The only standalone usage pattern in the reference: resetBuffer -> setOccupied(each pt) ->
    updateESDF3d (plan_manage/test/compare_topo.cpp:121-133)."""
    om = fo.OracleMap((6.0, 6.0, 3.0), optimistic=1)
    gm = fa.SDFMap((6.0, 6.0, 3.0), optimistic=1)
    rng = np.random.default_rng(2)
    pts = om.origin + np.array([6.0, 6.0, 3.0]) * rng.random((300, 3))
    pts = np.vstack([pts, [[50.0, 0, 0], [0, 0, -1.0]]])  # outside the map: ignored
    om.reset_buffer()
    gm.resetBuffer()
    for p in pts:
        om.set_occupied(p)
    gm.setOccupied(pts)
    om.update_esdf()
    gm.updateESDF3d()
    assert_map_equal(om, gm, helpers.full_box(om.nvox))
    lo, hi = (-1.0, -1.0, 0.0), (1.0, 1.5, 1.0)
    om.reset_buffer(lo, hi)
    gm.resetBuffer(lo, hi)
    assert_map_equal(om, gm, None, esdf=False)
    h = gm.syncHost(distance=True)
    assert np.abs(np.minimum(h["distance"], BIG) - np.minimum(om.dist, BIG)).max() <= ESDF_TOL
    gm.close()


def test_dist_grad_coarse_dist_and_state_queries(fa):
    om, _, _, box = helpers.explored_oracle_map((8.0, 6.0, 4.0), 12, 20)
    gm = gpu_twin(fa, om, box)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    rng = np.random.default_rng(0)
    pos = om.origin - 0.3 + (np.array([8.0, 6.0, 4.0]) + 0.6) * rng.random((5000, 3))
    d0, g0 = om.dist_grad(pos)
    d1, g1 = gm.getDistWithGrad(pos)
    assert np.abs(d0 - d1).max() <= ESDF_TOL and np.abs(g0 - g1).max() <= GRAD_TOL
    idx = rng.integers(-2, 85, size=(3000, 3)).astype(np.int32)
    o_g, i_g = gm.getOccupancy(idx)
    for k in range(0, 3000, 7):
        i3 = (fo.C.c_int * 3)(*idx[k])
        assert o_g[k] == om.L.fo_map_get_occupancy_idx(om.h, i3)
        assert i_g[k] == om.L.fo_map_get_inflate_idx(om.h, i3)
    dc = gm.getDistance(pos[:500])
    for k in range(0, 500, 5):
        i3 = (fo.C.c_int * 3)(*np.floor((pos[k] - om.origin) * 10).astype(int))
        assert abs(dc[k] - om.L.fo_map_get_distance_idx(om.h, i3)) <= ESDF_TOL
    gm.close()


def test_depth_insert_bit_exact_over_many_frames(fa):
    """inputPointCloud parity, including points outside the map, beyond max range, below z=0.2
    after clipping, duplicates in one end voxel, pcl-style 16-byte stride and an empty cloud."""
    map_size = (10.0, 8.0, 4.0)
    box = ((-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    om = fo.OracleMap(map_size, *box)
    gm = fa.SDFMap(map_size, *box)
    truth = om.fixture_world(3, 14)
    rng = np.random.default_rng(9)
    for k in range(30):
        pose = om.fixture_camera(truth, 5, k, 30, 0.9)
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2, maxdist=9.0 if k % 3 == 0 else 5.0)
        extra = pose[:3] + rng.normal(scale=6.0, size=(40, 3))  # some far outside the map
        pts = np.vstack([pts, extra.astype(np.float32), pts[:50]])
        om.input_points(pts, pose[:3])
        if k % 2:
            rec = np.zeros((len(pts), 4), dtype=np.float32)
            rec[:, :3] = pts
            from fuel_amd._lib import check
            check(gm.L.fuelmi_map_input_points(gm.h, rec.ctypes.data, 16, len(rec),
                                               (fo.C.c_double * 3)(*pose[:3])))
        else:
            gm.inputPointCloud(pts, pose[:3])
        assert om.get_local_bound() == gm.getLocalBound()
        assert np.array_equal(np.concatenate(om.get_updated_box()), np.concatenate(gm.getUpdatedBox()))
    gm.inputPointCloud(np.zeros((0, 3), np.float32), (0, 0, 0))  # reference: returns immediately
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    assert_map_equal(om, gm, om.get_local_bound())
    gm.close()


def test_frontier_incremental_rounds(fa):
    map_size = (20.0, 20.0, 5.0)
    org = (-10.0, -10.0, -1.0)
    box = ((org[0] + 1, org[1] + 1, 0.0), (9.0, 9.0, 3.0))
    om = fo.OracleMap(map_size, *box)
    gm = fa.SDFMap(map_size, *box)
    truth = om.fixture_world(42, 60)
    of = fo.OracleFrontier(om, 100)
    gf = fa.FrontierFinder(gm, cluster_min=100)
    k = 0
    for r in range(5):
        for _ in range(12):
            pose = om.fixture_camera(truth, 7, k, 60, 0.7)
            k += 1
            pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
            om.input_points(pts, pose[:3])
            gm.inputPointCloud(pts, pose[:3])
        n_o, n_g = of.search(), gf.searchFrontiers()
        assert n_o == n_g
        for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
            assert np.array_equal(a, b)  # frontier voxel indices bit-exact, cluster by cluster
        assert np.array_equal(of.flags, gf.flags())
        assert np.array_equal(of.removed_ids(), gf.removedIds())
        for c in range(n_o):
            for u, v in zip(of.cluster_info(0, c), gf.clusterInfo(0, c)):
                assert np.abs(u - v).max() < 1e-9
        of.commit(r == 2)
        gf.commit(r == 2)
        for which in (1, 2):
            for a, b in zip(sorted_clusters(of.clusters(which)), gf.clusters(which)):
                assert np.array_equal(a, b)
    gm.close()


def test_frontier_low_z_seeds_and_box_faces(fa):
    """Exploration box whose faces cut through frontier surfaces and a known region reaching below
    z = 0.4: exercises the non-qualified seeds (never added by BFS, but they start clusters)."""
    map_size = (8.0, 6.0, 4.0)
    box = ((-2.0, -1.5, -0.5), (1.0, 2.0, 1.0))
    om = fo.OracleMap(map_size, *box)
    truth = om.fixture_world(5, 6)
    om.fixture_known_state(truth, 5, 6, 1.0, 2.2)
    gm = gpu_twin(fa, om, box)
    for cmin in (0, 5, 60):
        of = fo.OracleFrontier(om, cmin)
        gf = fa.FrontierFinder(gm, cluster_min=cmin)
        om.set_updated_box((-1.0, -1.0, 0.2), (0.5, 1.0, 0.8))
        gm.setUpdatedBox((-1.0, -1.0, 0.2), (0.5, 1.0, 0.8))
        assert of.search() == gf.searchFrontiers()
        for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
            assert np.array_equal(a, b)
        assert np.array_equal(of.flags, gf.flags())
        gf.close()
    gm.close()


@pytest.mark.parametrize("cf_name", ["NORMAL|MINTIME", "NORMAL", "GUIDE_PHASE", "SMOOTH|WAYPT", "ALL"])
def test_bspline_cost_and_gradient(fa, cf_name):
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    gm = gpu_twin(fa, om, box)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    cf = {"NORMAL|MINTIME": fa.NORMAL_PHASE | fa.MINTIME, "NORMAL": fa.NORMAL_PHASE, "GUIDE_PHASE": fa.GUIDE_PHASE,
          "SMOOTH|WAYPT": fa.SMOOTHNESS | fa.WAYPOINTS, "ALL": 0x1FF}[cf_name]
    rng = np.random.default_rng(4)
    for Cn, N in [(64, 32), (3, 70), (5, 6)]:
        ctrl = helpers.make_trajectories(rng, Cn, N, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
        mint = bool(cf & fa.MINTIME)
        x, ptd, st, en = helpers.bspline_inputs(ctrl, 0.175, mint)
        guide = ctrl[:, 3:N - 3, :] + 0.1 if N > 6 else np.zeros((Cn, 0, 3))
        widx = np.array([1, N // 2, N - 3], dtype=np.int32)
        wp = ctrl[:, widx + 1, :] + 0.2
        wi = np.tile(widx, (Cn, 1))
        vpt = ctrl[:, N // 2, :] + 0.5
        vdir = np.tile(np.array([0.5, 1.0, 0.2]), (Cn, 1))
        vidx = np.full(Cn, N // 2 + 1, dtype=np.int32)
        opt = fa.BsplineOptimizer(ld_view=0.7)
        opt.setEnvironment(gm)
        pb = fa.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 3, 0.175, 1.0 if mint else None, guide, wp, wi,
                                    vpt, vdir, vidx)
        cg, gg = opt.combineCost(pb)
        for c in range(Cn):
            co, go = fo.bspline_cost_grad(om, x[c], N, cf, ptd[c], st[c], en[c], 3, 3, 0.175,
                                          1.0 if mint else -1.0, guide[c], wp[c], wi[c], (vpt[c], vdir[c], vidx[c]),
                                          ld_view=0.7)
            assert abs(cg[c] - co) <= 1e-6 * max(1.0, abs(co))
            assert np.abs(gg[c] - go).max() <= GRAD_TOL
    gm.close()


def test_bspline_one_dimensional_yaw_case(fa):
    """dim = 1 (yaw B-spline, planner_manager.cpp:774-865): order is forced to 3."""
    om = fo.OracleMap((4.0, 4.0, 2.0))
    gm = fa.SDFMap((4.0, 4.0, 2.0))
    rng = np.random.default_rng(8)
    N, Cn = 12, 4
    x = rng.normal(size=(Cn, N))
    cf = fa.SMOOTHNESS | fa.WAYPOINTS | fa.START | fa.END
    st = np.zeros((Cn, 3, 3))
    en = np.zeros((Cn, 3, 3))
    st[:, :, 0] = rng.normal(size=(Cn, 3))
    en[:, :, 0] = rng.normal(size=(Cn, 3))
    wp = np.zeros((Cn, 2, 3))
    wp[:, :, 0] = rng.normal(size=(Cn, 2))
    wi = np.tile(np.array([2, 6], np.int32), (Cn, 1))
    ptd = np.array([fo.bspline_pt_dist(x[c].reshape(N, 1)) for c in range(Cn)])
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    cg, gg = opt.combineCost(fa.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 1, 0.3, None, None, wp, wi))
    for c in range(Cn):
        co, go = fo.bspline_cost_grad(om, x[c], N, cf, ptd[c], st[c], en[c], 3, 1, 0.3, -1.0, None, wp[c], wi[c])
        assert abs(cg[c] - co) <= 1e-9 * max(1.0, abs(co)) and np.abs(gg[c] - go).max() <= 1e-8 * max(1, np.abs(go).max())
    # the whole yaw solve on the device (unbounded variables when dim = 1, :188-193) against the oracle's
    dev = opt.deviceProblem(fa.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 1, 0.3, None, None, wp, wi))
    # (run to convergence: the two runs agree to 5 digits for ~50 evaluations, then part ways with the
    # reduction order and meet again at the minimum; a cap in between would compare two mid-descent points)
    xs, cs, ev = dev.optimize(max_eval=2000)
    assert ev.max() < 2000
    for c in range(Cn):
        xo, co, eo = fo.bspline_optimize(om, x[c], N, cf, ptd[c], st[c], en[c], 3, 1, 0.3, -1.0, None, wp[c], wi[c],
                                         max_eval=2000)
        assert cs[c] < 0.01 * cg[c] and abs(cs[c] - co) <= 1e-3 * co, (c, cs[c], co, cg[c])
        chk, _ = fo.bspline_cost_grad(om, xs[c], N, cf, ptd[c], st[c], en[c], 3, 1, 0.3, -1.0, None, wp[c], wi[c])
        assert abs(chk - cs[c]) <= 1e-9 * max(1.0, abs(chk))
    gm.close()


def test_golden_fixture_replay(fa):
    """Replays the committed inputs of tests/golden/small_cycle.npz through the HIP path."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "small_cycle.npz"))
    gm = fa.SDFMap(mg.MAP_SIZE, *mg.BOX)
    for k in range(mg.N_FRAMES):
        gm.inputPointCloud(z["pts%d" % k], z["cam%d" % k])
    lo, hi = gm.getLocalBound()
    assert np.array_equal(np.array([lo, hi]), z["local_bound"])
    assert np.array_equal(np.concatenate(gm.getUpdatedBox()), z["updated_box"])
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    h = gm.syncHost(occupancy=True, inflate=True, distance=True)
    assert np.array_equal(h["occupancy"], z["occupancy"]) and np.array_equal(h["inflate"], z["inflate"])
    sl = tuple(slice(lo[i], hi[i] + 1) for i in range(3))
    assert np.abs(h["distance"].reshape(gm.nvox)[sl] - z["distance_box"]).max() <= ESDF_TOL
    gf = fa.FrontierFinder(gm, cluster_min=mg.CLUSTER_MIN)
    n = gf.searchFrontiers()
    off = z["cluster_offsets"]
    assert n == len(off) - 1
    for k, c in enumerate(gf.clusters(0)):
        assert np.array_equal(c, z["cluster_cells"][off[k]:off[k + 1]])
    assert np.array_equal(gf.flags(), z["frontier_flags"])
    ctrl, st, en = z["ctrl"], z["start"], z["end"]
    x = np.concatenate([ctrl.reshape(len(ctrl), -1), np.full((len(ctrl), 1), 0.2)], axis=1)
    ptd = np.array([fo.bspline_pt_dist(c) for c in ctrl])
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    cost, grad = opt.combineCost(fa.BsplineBatchProblem(x, ctrl.shape[1], fa.NORMAL_PHASE | fa.MINTIME, ptd, st,
                                                        en, 3, 3, 0.2))
    assert np.abs(cost - z["bspline_cost"]).max() <= 1e-6 * np.abs(z["bspline_cost"]).max()
    assert np.abs(grad - z["bspline_grad"]).max() <= GRAD_TOL
    d, g = gm.getDistWithGrad(z["query_pos"])
    assert np.abs(d - z["query_dist"]).max() <= ESDF_TOL and np.abs(g - z["query_grad"]).max() <= GRAD_TOL
    gm.close()


def test_full_size_g400_cycle_against_oracle_and_properties(fa):
    """BASELINE.json configs[1]/[2]: 400x400x100 @ 0.1 m, full-box cycle, 256 candidates."""
    import bench
    map_size, box, occ, ctrl64, _ = bench.build_inputs("G400", seed=42)
    om = fo.OracleMap(map_size, *box)
    om.occ[:] = occ
    gm = fa.SDFMap(map_size, *box)
    gm.uploadOccupancy(occ)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    h = assert_map_equal(om, gm, (lo, hi))
    # size-independent properties of an exact box-local EDT
    D = h["distance"].reshape(om.nvox)
    src = (h["inflate"].reshape(om.nvox) == 1) | (occ.reshape(om.nvox) < om.l_min - 1e-3)
    assert np.array_equal(D == 0.0, src)                      # zero exactly on the sources
    for ax in range(3):                                        # 1-Lipschitz along every axis
        assert np.abs(np.diff(D, axis=ax)).max() <= 0.1 + 1e-6
    q = np.round((D / 0.1) ** 2)                               # squared voxel distances are integers
    assert np.abs((D / 0.1) ** 2 - q).max() < 1e-3
    of = fo.OracleFrontier(om, 100)
    gf = fa.FrontierFinder(gm, cluster_min=100)
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    assert of.search() == gf.searchFrontiers()
    for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
        assert np.array_equal(a, b)
    assert np.array_equal(of.flags, gf.flags())
    # idempotence: nothing changed -> nothing new, nothing removed
    gf.commit()
    gm.setUpdatedBox(*box)
    assert gf.searchFrontiers() == 0 and len(gf.removedIds()) == 0
    rng = np.random.default_rng(77)
    ctrl = bench.make_trajectories(rng, 256, 32, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
    x, ptd, st, en = bench.bspline_problem(ctrl, 0.175)
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    cf = fa.NORMAL_PHASE | fa.MINTIME
    cg, gg = opt.combineCost(fa.BsplineBatchProblem(x, 32, cf, ptd, st, en, 3, 3, 0.175))
    for c in range(0, 256, 3):
        co, go = fo.bspline_cost_grad(om, x[c], 32, cf, ptd[c], st[c], en[c], 3, 3, 0.175)
        assert abs(cg[c] - co) <= 1e-6 * max(1.0, abs(co)) and np.abs(gg[c] - go).max() <= GRAD_TOL
    gm.close()


def test_g800_streaming_inserts_incremental_esdf_and_frontier(fa):
    """BASELINE.json configs[3]: 800x800x200 @ 0.1 m, streaming depth frames -> fusion ->
    box-local inflate + ESDF -> incremental frontier search, frame by frame against the oracle."""
    map_size = (80.0, 80.0, 20.0)
    box = ((-39.0, -39.0, 0.0), (39.0, 39.0, 15.0))
    om = fo.OracleMap(map_size, *box)
    gm = fa.SDFMap(map_size, *box)
    assert gm.nvox == (800, 800, 200)
    truth = om.fixture_world(42, 3200)
    of = fo.OracleFrontier(om, 100)
    gf = fa.FrontierFinder(gm, cluster_min=100)
    n_frames = 16
    touched_lo = np.array(om.nvox)
    touched_hi = np.zeros(3, dtype=int)
    for k in range(n_frames):
        pose = om.fixture_camera(truth, 7, k, 400, 0.3)  # a short stretch of the tour
        pts = om.fixture_render(truth, pose, 320, 240, 2, 2)
        om.input_points(pts, pose[:3])
        gm.inputPointCloud(pts, pose[:3])
        lo, hi = om.get_local_bound()
        assert (lo, hi) == gm.getLocalBound()
        om.inflate_local()
        om.update_esdf()
        gm.clearAndInflateLocalMap()
        gm.updateESDF3d()
        touched_lo = np.minimum(touched_lo, lo)
        touched_hi = np.maximum(touched_hi, hi)
        if k % 4 == 3:
            n_o, n_g = of.search(), gf.searchFrontiers()
            assert n_o == n_g
            for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
                assert np.array_equal(a, b)
            assert np.array_equal(of.removed_ids(), gf.removedIds())
            of.commit()
            gf.commit()
    sl = tuple(slice(int(touched_lo[i]), int(touched_hi[i]) + 1) for i in range(3))
    bx = (tuple(int(v) for v in touched_lo), tuple(int(v) for v in touched_hi))
    h = gm.syncHost(occupancy=True, inflate=True, distance=True, box=bx)
    for name, ref_arr in (("occupancy", om.occ), ("inflate", om.infl)):
        assert np.array_equal(h[name].reshape(om.nvox)[sl], ref_arr.reshape(om.nvox)[sl]), name
    assert np.abs(np.clip(h["distance"].reshape(om.nvox)[sl], -BIG, BIG) -
                  np.clip(om.dist.reshape(om.nvox)[sl], -BIG, BIG)).max() <= ESDF_TOL
    assert np.array_equal(of.flags, gf.flags())
    gm.close()


def test_frontier_many_small_clusters_two_radix_passes(fa):
    """cluster_min = 0 keeps every cluster (hundreds to thousands of tiny ones): exercises the second
    radix pass of the device grouping (> 256 clusters), the > 512-record download and the launch
    re-estimation path."""
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    rng = np.random.default_rng(11)
    # sprinkle isolated unknown voxels into free space: each creates a little frontier shell
    occ = om.occ.reshape(om.nvox)
    free = np.argwhere((occ >= om.l_min - 1e-3) & (occ <= om.l_occ))
    pick = free[np.all(free % 5 == 2, axis=1)]  # a lattice: the shells stay separate clusters
    occ[tuple(pick.T)] = om.l_min - 0.01
    gm = gpu_twin(fa, om, box)
    for cmin in (0, 3):
        of = fo.OracleFrontier(om, cmin)
        gf = fa.FrontierFinder(gm, cluster_min=cmin)
        for rnd in range(2):  # second round: estimates from the first search are reused
            om.set_updated_box(*box)
            gm.setUpdatedBox(*box)
            n_o, n_g = of.search(), gf.searchFrontiers()
            assert n_o == n_g
            if rnd == 0:
                assert n_o > 512
            for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
                assert np.array_equal(a, b)
            assert np.array_equal(of.flags, gf.flags())
            for c in range(0, n_o, 37):
                for u, v in zip(of.cluster_info(0, c), gf.clusterInfo(0, c)):
                    assert np.abs(u - v).max() < 1e-9
            of.commit()
            gf.commit()
        gf.close()
    gm.close()


# ------------------------------------------------------------------------------------------------
# depth projection on the device (MapROS::proessDepthImage, map_ros.cpp:176-215)
# ------------------------------------------------------------------------------------------------
def _depth_frames(w, truth, n, width, height, seed=3):
    from fuel_amd import synth
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        pose = w.camera(truth, 5, k, n, 0.9)
        img = w.depth_image(truth, pose, width, height, max_range=7.0)
        img[rng.random(img.shape) < 0.02] = 0
        img[rng.random(img.shape) < 0.01] = rng.integers(1, 199)
        out.append((img, pose, synth.World.pose_quaternion(pose)))
    return out


def _scaled_cfg(mod, width, **kw):
    s = width / 640.0
    return mod(fx=387.229248046875 * s, fy=387.229248046875 * s, cx=321.04638671875 * s,
               cy=243.44969177246094 * s, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("margin,skip", [(2, 2), (0, 3), (1, 1), (5, 4)])
def test_depth_projection_bit_exact(fa, margin, skip):
    """Projected float points identical to the oracle's (same order), incl. margin 0 where the zero test
    runs into the next row and, on the last row, past the image (defined as 0)."""
    from fuel_amd import synth
    map_size = (10.0, 8.0, 4.0)
    gm = fa.SDFMap(map_size, (-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    w = synth.World.for_map_size(map_size)
    truth = w.world(3, 14)
    for img, pose, q in _depth_frames(w, truth, 5, 200, 150):
        a = fo.project_depth(img, pose[:3], q, _scaled_cfg(fo.depth_cfg, 200, margin=margin, skip=skip))
        b = gm.projectDepthImage(img, pose[:3], q, _scaled_cfg(gm.depthConfig, 200, margin=margin, skip=skip))
        assert a.shape == b.shape and np.array_equal(a, b)
    gm.close()


@pytest.mark.gpu
def test_depth_image_fusion_matches_projection_plus_insert(fa):
    """fuelmi_map_input_depth == oracle projection followed by the oracle's inputPointCloud: bit-exact
    log-odds, bounds and update box over a frame sequence; frames from outside the map and frames
    whose pixels are all filtered out leave the map (and raycast_num_) untouched."""
    from fuel_amd import synth
    map_size = (10.0, 8.0, 4.0)
    box = ((-4.0, -3.0, 0.0), (4.0, 3.0, 2.2))
    om = fo.OracleMap(map_size, *box)
    gm = fa.SDFMap(map_size, *box)
    w = synth.World.for_map_size(map_size)
    truth = w.world(3, 14)
    ocfg, gcfg = _scaled_cfg(fo.depth_cfg, 160), _scaled_cfg(gm.depthConfig, 160)
    frames = _depth_frames(w, truth, 20, 160, 120)
    for k, (img, pose, q) in enumerate(frames):
        if k == 7:  # every pixel closer than depth_filter_mindist: proj_points_cnt = 0 -> early return
            near = np.full_like(img, 100)
            assert gm.inputDepthImage(near, pose[:3], q, gcfg) == 0
            assert len(fo.project_depth(near, pose[:3], q, ocfg)) == 0
        if k == 11:  # camera outside the map: depthPoseCallback returns before projecting
            assert gm.inputDepthImage(img, (50.0, 0.0, 1.0), q, gcfg) == 0
        pts = fo.project_depth(img, pose[:3], q, ocfg)
        om.input_points(pts, pose[:3])
        assert gm.inputDepthImage(img, pose[:3], q, gcfg) == len(pts)
        assert om.get_local_bound() == gm.getLocalBound()
        assert np.array_equal(np.concatenate(om.get_updated_box()), np.concatenate(gm.getUpdatedBox()))
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    assert_map_equal(om, gm, om.get_local_bound())
    gm.close()


# ------------------------------------------------------------------------------------------------
# splitLargeFrontiers + down-sampling on the device
# ------------------------------------------------------------------------------------------------
def _assert_split_equal(of, gf, om, n):
    """Same pieces in the same order; cells as sets (the oracle keeps BFS order, the device address
    order); mean/AABB to 1e-9; filtered cells bit-equal as leaf-ordered float lists (the oracle runs
    with canonical_order: the VoxelGrid's float sums then see the cells in address order on both sides)."""
    ca, cb = of.clusters(0), gf.clusters(0)
    assert len(ca) == len(cb) == n
    for k in range(n):
        assert np.array_equal(np.sort(ca[k]), cb[k]), "piece %d differs" % k
        ia, ib = of.cluster_info(0, k), gf.clusterInfo(0, k)
        for x, y in zip(ia, ib):
            assert np.array_equal(np.asarray(x), np.asarray(y))  # order-free means: bit-equal
        fa_, fb_ = of.filtered(0, k), gf.filtered(0, k)
        assert fa_.shape == fb_.shape and len(fa_) > 0
        assert np.array_equal(fa_.astype(np.float32), fb_)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,size_xy", [(42, 2.0), (7, 1.2), (11, 3.0)])
def test_split_large_frontiers_matches_oracle(fa, seed, size_xy):
    om, truth, frames, box = helpers.explored_oracle_map((16.0, 14.0, 4.0), 30, 28, seed=seed)
    gm = gpu_twin(fa, om, box)
    ub = om.get_updated_box(reset=False)
    gm.setUpdatedBox(*ub)
    of = fo.OracleFrontier(om, cluster_min=60, cluster_size_xy=size_xy, down_sample=3, split=True,
                           canonical_order=True)
    gf = fa.FrontierFinder(gm, cluster_min=60, cluster_size_xy=size_xy, down_sample=3, split=True)
    plain = fo.OracleFrontier(om, cluster_min=60)
    n0 = plain.search()
    om.set_updated_box(*ub)
    n1 = of.search()
    n2 = gf.searchFrontiers()
    assert n1 == n2 and n1 > n0 > 0
    _assert_split_equal(of, gf, om, n1)
    # committed pieces survive as ordinary clusters (cells materialised in address order)
    gf.commit()
    of.commit()
    for a, b in zip(of.clusters(1), gf.clusters(1)):
        assert np.array_equal(np.sort(a), b)
    gf.close()
    gm.close()


@pytest.mark.gpu
def test_split_with_low_z_seed_clusters(fa):
    """Pieces of clusters that were started by a seed below min_z (an NQ seed: part of the cluster,
    never grown from) keep that seed cell through the split."""
    om, truth, frames, box = helpers.explored_oracle_map((12.0, 12.0, 4.0), 16, 22, seed=5, extent=0.8)
    gm = gpu_twin(fa, om, box)
    ub = om.get_updated_box(reset=False)
    gm.setUpdatedBox(*ub)
    of = fo.OracleFrontier(om, cluster_min=20, cluster_size_xy=1.0, down_sample=3, split=True, canonical_order=True)
    gf = fa.FrontierFinder(gm, cluster_min=20, cluster_size_xy=1.0, down_sample=3, split=True)
    n1, n2 = of.search(), gf.searchFrontiers()
    assert n1 == n2 > 0
    _assert_split_equal(of, gf, om, n1)
    gf.close()
    gm.close()


# ------------------------------------------------------------------------------------------------
# viewpoint sampling / coverage on the device
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seed,size_xy", [(42, 2.0), (7, 1.2)])
def test_viewpoint_sampling_and_coverage_match_oracle(fa, seed, size_xy):
    """computeFrontiersToVisit: same active/dormant partition, per cluster the same viewpoints in the same
    order with identical coverage counts; positions to 1e-9 m (cluster means come from exact integer
    sums on the device, from a sequential f64 sum in the reference), yaws to 1e-9 rad (device libm, tree
    sum).  Then isFrontierCovered after more fusion, step by step."""
    om, truth, frames, box = helpers.explored_oracle_map((16.0, 14.0, 4.0), 30, 28, seed=seed)
    om.inflate_local()
    gm = gpu_twin(fa, om, box)
    gm.setLocalBound(*om.get_local_bound())
    gm.clearAndInflateLocalMap()
    ub = om.get_updated_box(reset=False)
    gm.setUpdatedBox(*ub)
    of = fo.OracleFrontier(om, cluster_min=60, cluster_size_xy=size_xy, down_sample=3, split=True,
                           canonical_order=True)
    of.set_viewpoint_cfg(fo.viewpoint_cfg(min_visib_num=5))
    gf = fa.FrontierFinder(gm, cluster_min=60, cluster_size_xy=size_xy, down_sample=3, split=True)
    gf.setViewpointConfig(gf.viewpointConfig(min_visib_num=5))
    assert of.search() == gf.searchFrontiers() > 0
    of.compute_to_visit()
    na, nd = gf.computeFrontiersToVisit()
    assert na == len(of.clusters(1)) > 0 and nd == len(of.clusters(2))
    total = 0
    for k in range(na):
        (pa, va), (pb, vb) = of.viewpoints(1, k), gf.viewpoints(1, k)
        assert np.array_equal(va, vb), "coverage counts of cluster %d differ" % k
        assert np.array_equal(pa[:, :3], pb[:, :3])
        dyaw = np.abs(pa[:, 3] - pb[:, 3])
        assert np.minimum(dyaw, 2 * np.pi - dyaw).max() <= 1e-9
        total += len(va)
    assert total > 20
    for which in (1, 2):
        for a, b in zip(of.clusters(which), gf.clusters(which)):
            assert np.array_equal(np.sort(a), b)
    # getTopViewpointsInfo mirror: one entry per active frontier
    pts, yaws, avgs = gf.getTopViewpointsInfo((0.0, 0.0, 1.0))
    assert len(pts) == na
    # coverage check while the map keeps changing
    assert of.is_covered() == gf.isFrontierCovered()
    seen = 0
    for k in range(6):
        pose = om.fixture_camera(truth, 99, k, 6, 0.6)
        pts_ = om.fixture_render(truth, pose, 160, 120, 2, 2)
        om.input_points(pts_, pose[:3])
        gm.inputPointCloud(pts_, pose[:3])
        a, b = of.is_covered(), gf.isFrontierCovered()
        assert a == b
        seen += int(a)
    assert seen > 0
    gf.close()
    gm.close()


# ------------------------------------------------------------------------------------------------
# whole trajectory solves on the device
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_bspline_device_optimizer_against_oracle_lbfgs(fa):
    """fuelmi_bspline_dev_optimize vs the oracle's fo_bspline_optimize (the same box-projected L-BFGS,
    sequential f64): every candidate ends at a cost no worse than 0.1 % above the oracle's and far
    below its start; bounds hold (exploration box shrunk by 0.1, knot span in [0,5]); the evaluation cap
    is honoured; the returned variables reproduce the returned cost; a candidate solved alone gives the
    same answer as inside a batch."""
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    gm = gpu_twin(fa, om, box)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    cf = fa.NORMAL_PHASE | fa.MINTIME
    rng = np.random.default_rng(12)
    Cn, N, dt = 24, 32, 0.175
    ctrl = helpers.make_trajectories(rng, Cn, N, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
    x, ptd, st, en = helpers.bspline_inputs(ctrl, dt, True)
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    pb = fa.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 3, dt)
    dev = opt.deviceProblem(pb)
    c0, _ = opt.combineCost(pb)
    xg, cg, eg = dev.optimize(max_eval=300)
    assert eg.max() <= 300 and eg.min() > 5
    blo, bhi = np.array(box[0]) + 0.1, np.array(box[1]) - 0.1
    pts = xg[:, :-1].reshape(Cn, N, 3)
    assert (pts >= blo - 1e-12).all() and (pts <= bhi + 1e-12).all()
    assert (xg[:, -1] >= 0.0).all() and (xg[:, -1] <= 5.0).all()
    worse = 0
    for c in range(Cn):
        xo, fo_cost, evo = fo.bspline_optimize(om, x[c], N, cf, ptd[c], st[c], en[c], 3, 3, dt, max_eval=300)
        assert cg[c] < 0.5 * c0[c] and fo_cost < 0.5 * c0[c]      # both actually optimise
        chk, _ = fo.bspline_cost_grad(om, xg[c], N, cf, ptd[c], st[c], en[c], 3, 3, dt)
        assert abs(chk - cg[c]) <= 1e-6 * max(1.0, abs(chk))      # returned x <-> returned cost
        if cg[c] > fo_cost * 1.001 + 1e-9:
            worse += 1
    assert worse <= Cn // 8  # reduction order may send a few solves down another line-search branch
    # a capped run stops at the cap and is no better than the long one
    xg2, cg2, eg2 = dev.optimize(max_eval=20)
    assert eg2.max() <= 20 and (cg2 >= cg - 1e-9).all()
    # solving candidate 3 alone == inside the batch
    pb1 = fa.BsplineBatchProblem(x[3:4], N, cf, ptd[3:4], st[3:4], en[3:4], 3, 3, dt)
    x1, c1, e1 = opt.deviceProblem(pb1).optimize(max_eval=300)
    assert np.array_equal(x1[0], xg[3]) and c1[0] == cg[3] and e1[0] == eg[3]
    gm.close()


def _sample_paths(rng, cn, k, lo, hi):
    """cn smooth sample sets inside [lo, hi] + start/end velocity and acceleration, as getSamples() hands them over."""
    pts = np.empty((cn, k, 3))
    for c in range(cn):
        a = lo + (hi - lo) * rng.random(3)
        b = lo + (hi - lo) * rng.random(3)
        s = np.linspace(0, 1, k)[:, None]
        pts[c] = a + (b - a) * s + 0.4 * np.sin(np.pi * s * rng.integers(1, 4)) * rng.normal(size=3)
    pts = np.clip(pts, lo, hi)
    der = rng.normal(scale=0.7, size=(cn, 4, 3))
    return pts, der


@pytest.mark.gpu
@pytest.mark.parametrize("degree", [3, 4, 5])
def test_spline_parameterize_and_boundary_states_match_oracle(fa, degree):
    """fuelmi_bspline_parameterize / _boundary_states vs the oracle's restatement of NonUniformBspline
    (itself pinned on the real class): control points within 1e-9 (the reference solves by QR, the device by
    banded normal equations + one refinement step), boundary states within 1e-10 (same arithmetic, FMA
    contraction aside).  K = 2 (the smallest the reference accepts), ragged knot spans, 150 samples."""
    gm = fa.SDFMap((8.0, 8.0, 3.0))
    rng = np.random.default_rng(40 + degree)
    for cn, k in [(1, 2), (5, 3), (33, 17), (3, 150)]:
        ts = 0.1 + 0.4 * rng.random(cn)
        pts, der = _sample_paths(rng, cn, k, np.array([-3.0, -3.0, 0.2]), np.array([3.0, 3.0, 1.8]))
        got = fa.NonUniformBspline.parameterizeToBspline(gm, ts, pts, der, degree)
        assert got.shape == (cn, k + degree - 1, 3)
        for c in range(cn):
            want = fo.spline_parameterize(ts[c], pts[c], der[c], degree)
            assert np.abs(got[c] - want).max() <= 1e-9, (cn, k, c)
        for ks, ke in [(2, 0), (2, 2), (degree, 1), (0, 0)]:
            s, e = fa.NonUniformBspline.getBoundaryStates(gm, got, ts, degree, ks, ke)
            for c in range(cn):
                s0, e0 = fo.spline_boundary_states(got[c], ts[c], degree, ks, ke)
                scale = max(1.0, np.abs(s0).max(), np.abs(e0).max())
                assert np.abs(s[c] - s0).max() <= 1e-10 * scale and np.abs(e[c] - e0).max() <= 1e-10 * scale
    with pytest.raises(fa.FuelmiError):  # "[B-spline]:time step error."
        fa.NonUniformBspline.parameterizeToBspline(gm, np.array([0.0]), pts[:1], der[:1], degree)
    with pytest.raises(fa.FuelmiError):  # "point set have only 1 points"
        fa.NonUniformBspline.parameterizeToBspline(gm, np.array([0.2]), pts[:1, :1], der[:1], degree)
    with pytest.raises(fa.FuelmiError):
        fa.NonUniformBspline.parameterizeToBspline(gm, np.array([0.2]), pts[:1], der[:1], 6)
    gm.close()


@pytest.mark.gpu
def test_samples_to_solve_on_the_device_matches_the_stepwise_oracle(fa):
    """planner_manager.cpp:161-184 as one device sequence (loadSamples -> eval / optimize) against the same
    steps done one by one with the oracle: parameterizeToBspline -> getBoundaryStates(2,0) -> pt_dist_ ->
    combineCost.  Cost rel 1e-6, gradient 1e-4 (the §8d bar); the solve then improves every candidate."""
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    gm = gpu_twin(fa, om, box)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    cf = fa.NORMAL_PHASE | fa.MINTIME
    rng = np.random.default_rng(77)
    cn, k, degree = 20, 30, 3
    n = k + degree - 1
    ts = 0.15 + 0.1 * rng.random(cn)
    pts, der = _sample_paths(rng, cn, k, np.array(box[0]) + 0.6, np.array(box[1]) - 0.6)
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    # the batch is created with placeholders; loadSamples replaces x, knot span, pt_dist and the boundary states
    x0 = np.zeros((cn, 3 * n + 1))
    x0[:, -1] = 1.0
    pb = fa.BsplineBatchProblem(x0, n, cf, np.ones(cn), np.zeros((cn, 3, 3)), np.zeros((cn, 3, 3)), 1, 3, 1.0)
    dev = opt.deviceProblem(pb)
    dev.loadSamples(ts, pts, der)
    dev.eval()
    cost, grad = dev.download()
    f0 = np.empty(cn)
    for c in range(cn):
        ctrl = fo.spline_parameterize(ts[c], pts[c], der[c], degree)
        st, en = fo.spline_boundary_states(ctrl, ts[c], degree, 2, 0)
        en3 = np.zeros((3, 3))
        en3[0] = en[0]
        x = np.concatenate([ctrl.reshape(-1), [ts[c]]])
        f, g = fo.bspline_cost_grad(om, x, n, cf, fo.bspline_pt_dist(ctrl), st, en3, 1, 3, ts[c])
        f0[c] = f
        assert abs(cost[c] - f) <= 1e-6 * max(1.0, abs(f)), c
        assert np.abs(grad[c] - g).max() <= GRAD_TOL * max(1.0, np.abs(g).max()), c
    xg, cg, eg = dev.optimize(max_eval=200)
    assert (cg <= f0 + 1e-9).all() and (cg < 0.9 * f0).sum() >= cn // 2
    # a second load into the same batch (the next replan) starts from the new samples, not the old solution
    dev.loadSamples(ts[::-1].copy(), pts[::-1].copy(), der[::-1].copy())
    dev.eval()
    cost2, _ = dev.download()
    assert np.allclose(cost2, cost[::-1], rtol=1e-9, atol=1e-12)
    gm.close()


@pytest.mark.gpu
def test_frontier_regrows_dropped_clusters_beyond_the_scan_box(fa):
    """The search only processes the region that can hold new cells (scan box + boxes of the clusters it
    just dropped).  A large cluster clipped by a small update is dropped as a whole and must be re-grown
    over its full extent, far outside the scan box; untracked changes (uploadOccupancy) reopen the whole
    box.  Frame-by-frame against the oracle, flags included."""
    map_size = (24.0, 24.0, 4.0)
    box = ((-11.0, -11.0, 0.0), (11.0, 11.0, 2.2))
    om = fo.OracleMap(map_size, *box)
    gm = fa.SDFMap(map_size, *box)
    truth = om.fixture_world(9, 40)
    of = fo.OracleFrontier(om, 40)
    gf = fa.FrontierFinder(gm, cluster_min=40)
    outside = 0
    for k in range(40):
        pose = om.fixture_camera(truth, 3, k, 40, 0.75)
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
        om.input_points(pts, pose[:3])
        gm.inputPointCloud(pts, pose[:3])
        ub = om.get_updated_box(reset=False)
        before = [(of.cluster_info(1, c)[1], of.cluster_info(1, c)[2]) for c in range(len(of.clusters(1)))]
        n_o, n_g = of.search(), gf.searchFrontiers()
        assert n_o == n_g
        for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
            assert np.array_equal(a, b)
        assert np.array_equal(of.flags, gf.flags())
        rid = list(of.removed_ids())
        assert rid == list(gf.removedIds())
        # did a dropped cluster reach more than 1.5 m beyond the scan box (updated box +- 1 m)?
        alive = list(range(len(before)))
        for r in rid:
            lo, hi = before[alive.pop(r)]
            if (lo < np.array(ub[0]) - 2.5).any() or (hi > np.array(ub[1]) + 2.5).any():
                outside += 1
        of.commit()
        gf.commit()
    assert outside >= 3
    # untracked change: the whole occupancy is replaced behind the finder's back
    gm.uploadOccupancy(om.occ)
    of2 = fo.OracleFrontier(om, 40)
    gf.reset()
    om.set_updated_box((-1.0, -1.0, 0.5), (1.0, 1.0, 1.5))
    gm.setUpdatedBox((-1.0, -1.0, 0.5), (1.0, 1.0, 1.5))
    assert of2.search() == gf.searchFrontiers()
    for a, b in zip(sorted_clusters(of2.clusters(0)), gf.clusters(0)):
        assert np.array_equal(a, b)
    assert np.array_equal(of2.flags, gf.flags())
    gm.close()


@pytest.mark.gpu
def test_error_paths_of_the_front_end_calls(fa):
    """Misuse is reported through the status code / FuelmiError, never by a crash or a silent fallback."""
    from fuel_amd._lib import FuelmiError
    gm = fa.SDFMap((6.0, 6.0, 3.0))
    gf = fa.FrontierFinder(gm, cluster_min=10)                      # split off: no filtered cells
    with pytest.raises(FuelmiError):
        gf.computeFrontiersToVisit()                                # no viewpoint configuration yet
    with pytest.raises(FuelmiError):
        gf.setViewpointConfig(gf.viewpointConfig(rnum=0))           # invalid configuration
    gf.setViewpointConfig(gf.viewpointConfig())
    with pytest.raises(FuelmiError):
        gf.searchFrontiersEnd()                                     # no matching _begin
    assert gf.isFrontierCovered() is False                          # nothing to cover yet
    # between _begin and _end the cluster lists belong to the search: modifying calls are refused
    gf.searchFrontiersBegin()
    for call in (gf.searchFrontiersBegin, gf.commit, gf.reset, gf.computeFrontiersToVisit, gf.isFrontierCovered):
        with pytest.raises(FuelmiError):
            call()
    gf.searchFrontiersEnd()
    gf.commit()
    with pytest.raises(FuelmiError):
        gm.inputDepthImage(np.zeros((8, 8), np.uint16), (0, 0, 1), (1, 0, 0, 0), gm.depthConfig(skip=0))
    # a frame from outside the map and an all-too-near frame are ignored, not errors
    assert gm.inputDepthImage(np.full((48, 64), 1500, np.uint16), (100.0, 0.0, 1.0), (1, 0, 0, 0)) == 0
    assert gm.inputDepthImage(np.full((48, 64), 50, np.uint16), (0.0, 0.0, 1.0), (1, 0, 0, 0)) == 0
    ctrl = np.zeros((1, 8, 3))
    ctrl[0, :, 0] = np.linspace(-1, 1, 8)
    x = np.concatenate([ctrl.reshape(1, -1), [[0.2]]], axis=1)
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    pb = fa.BsplineBatchProblem(x, 8, fa.NORMAL_PHASE | fa.MINTIME, np.array([0.25]), np.zeros((1, 3, 3)),
                                np.zeros((1, 3, 3)), 3, 3, 0.2)
    dev = opt.deviceProblem(pb)
    with pytest.raises(FuelmiError):
        dev.optimize(max_eval=0)
    xo, co, ev = dev.optimize(max_eval=1)  # the cap is honoured down to a single evaluation
    assert ev[0] == 1 and np.isfinite(co[0])
    gf.close()
    gm.close()


@pytest.mark.gpu
def test_esdf_on_a_very_long_x_line(fa):
    """x lines of 1300 voxels: the x pass switches to its 16-column tile (the 32-column one would not fit
    the LDS); result against the oracle."""
    map_size = (130.0, 4.0, 4.0)
    om = fo.OracleMap(map_size)
    rng = np.random.default_rng(21)
    occ = om.occ.reshape(om.nvox)
    occ[:] = om.l_min                                            # everything known free
    idx = rng.integers(0, [om.nvox[0], om.nvox[1], om.nvox[2]], size=(60, 3))
    occ[idx[:, 0], idx[:, 1], idx[:, 2]] = om.l_occ + 0.5         # a few obstacles far apart along x
    occ[900:, :, :] = om.unknown_value                            # and an unknown end
    gm = fa.SDFMap(map_size)
    gm.uploadOccupancy(om.occ)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    assert_map_equal(om, gm)
    gm.close()
