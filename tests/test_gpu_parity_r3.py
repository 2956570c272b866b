"""Round-3 parity cases through the C-ABI against the oracle.

* BASELINE config #4 AT ITS OWN SIZE: the 640x480 depth frames bench.py times on the 800x800x200 map (skip 2,
  ~19 k rays per frame), read in place from device memory by fuelmi_map_input_depth -- the four-lanes-per-ray
  split, the 64x64x32 LDS fan bitmap overflowing into global atomics, the region-restricted incremental search --
  frame by frame against the oracle, then once more through the C++ loop bench.py times (fuelmi_bench_stream).
  Reference path: plan_env/src/map_ros.cpp:121-150,176-215, sdf_map.cpp:259-345.
* the frontier chain on inputs chosen against its tile machinery: regions not aligned to tiles, NQ seeds on the
  box_max face, many clusters per tile column, both orders of the grouped output."""
import numpy as np
import pytest

import helpers
from oracle import fuel_oracle as fo

pytestmark = pytest.mark.gpu

ESDF_TOL = 1e-4
BIG = 1e6


@pytest.fixture(scope="module")
def fa():
    import fuel_amd
    assert fuel_amd.lib().fuelmi_device_count() > 0, "no GPU visible: the HIP path cannot run"
    return fuel_amd


def sorted_clusters(cl):
    return [np.sort(c) for c in cl]


def test_config4_streaming_at_its_own_size(fa):
    import bench
    map_size, n_obs, _ = bench.WORKLOADS["G800S"]
    box = bench.exploration_box(map_size)
    n_frames = 14
    frames = bench.streaming_frames(map_size, n_obs, n_frames, seed=42)
    assert frames[0][0].shape == (480, 640)
    rng = np.random.default_rng(1042)
    ctrl = bench.make_trajectories(rng, 8, 32, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
    om = fo.OracleMap(map_size, *box)
    of = fo.OracleFrontier(om, 100)
    cyc = bench.GpuStreamCycle(map_size, box, frames, ctrl, device=0)
    gm, gf = cyc.map, cyc.ff
    assert gm.nvox == (800, 800, 200)
    touched_lo, touched_hi = np.array(om.nvox), np.zeros(3, dtype=int)
    n_rays = []
    for i, (img, pos, q) in enumerate(frames):
        pts = fo.project_depth(img, pos, q)
        n_g = cyc.fuse(i)  # device-resident frame, read in place
        assert n_g == len(pts), "frame %d: %d projected points vs %d" % (i, n_g, len(pts))
        n_rays.append(n_g)
        gf.searchFrontiersBegin()
        if len(pts):
            om.input_points(pts, pos)
            lo, hi = om.get_local_bound()
            assert (lo, hi) == gm.getLocalBound(), "frame %d: local bound" % i
            om.inflate_local()
            om.update_esdf()
            gm.clearAndInflateLocalMap()
            gm.updateESDF3d()
            touched_lo, touched_hi = np.minimum(touched_lo, lo), np.maximum(touched_hi, hi)
            # log-odds of the frame's box, bit for bit, every frame
            bx = (tuple(int(v) for v in lo), tuple(int(v) for v in hi))
            sl = tuple(slice(bx[0][k], bx[1][k] + 1) for k in range(3))
            h = gm.syncHost(occupancy=True, box=bx)
            assert np.array_equal(h["occupancy"].reshape(om.nvox)[sl], om.occ.reshape(om.nvox)[sl]), "frame %d: log-odds" % i
        n_o = of.search()
        n_n = gf.searchFrontiersEnd()
        assert n_o == n_n, "frame %d: %d new clusters vs %d" % (i, n_n, n_o)
        for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
            assert np.array_equal(a, b), "frame %d: cluster cells" % i
        assert np.array_equal(of.removed_ids(), gf.removedIds()), "frame %d: removed_ids_" % i
        of.commit()
        gf.commit()
        if i in (4, n_frames - 1):
            assert np.array_equal(of.flags, gf.flags()), "frame %d: frontier_flag_" % i
    assert max(n_rays) > 10000, "the frames are supposed to be ~19 k rays each (%s)" % n_rays
    bx = (tuple(int(v) for v in touched_lo), tuple(int(v) for v in touched_hi))
    sl = tuple(slice(bx[0][k], bx[1][k] + 1) for k in range(3))
    h = gm.syncHost(occupancy=True, inflate=True, distance=True, box=bx)
    for name, ref_arr in (("occupancy", om.occ), ("inflate", om.infl)):
        assert np.array_equal(h[name].reshape(om.nvox)[sl], ref_arr.reshape(om.nvox)[sl]), name
    assert np.abs(np.clip(h["distance"].reshape(om.nvox)[sl], -BIG, BIG) -
                  np.clip(om.dist.reshape(om.nvox)[sl], -BIG, BIG)).max() <= ESDF_TOL
    committed = [c.copy() for c in gf.clusters(1)]
    assert len(committed) == len(of.clusters(1))
    for a, b in zip(sorted_clusters(of.clusters(1)), committed):
        assert np.array_equal(a, b)
    occ_box = h["occupancy"].reshape(om.nvox)[sl].copy()
    cyc.close()

    # the same frames through the C++ loop bench.py times: same map, same committed clusters
    cyc2 = bench.GpuStreamCycle(map_size, box, frames, ctrl, device=0)
    cyc2.run_native(n_frames)
    cyc2.finish()
    h2 = cyc2.map.syncHost(occupancy=True, box=bx)
    assert np.array_equal(h2["occupancy"].reshape(om.nvox)[sl], occ_box)
    c2 = [c.copy() for c in cyc2.ff.clusters(1)]
    assert len(c2) == len(committed) and all(np.array_equal(x, y) for x, y in zip(c2, committed))
    fs = cyc2.ff.stats()
    assert fs[0] > 0 and fs[2] == 0, "the searches are supposed to run on the fast chain (%s)" % (fs,)
    cyc2.close()


def test_esdf_kernel_choice_follows_the_place_not_the_previous_update(fa, monkeypatch):
    """The local bound alternates between an explored hall (only the floor and a pillar are sources: outputs tens of
    voxels from any source -> far-field kernels) and a half-explored region (unknown space is a source everywhere ->
    plain kernels).  The statistic is kept per group of x-slabs, so from the second visit on each place gets its own
    kernels -- the choice of round 2 ("whatever the previous update saw") paid the wrong family on every call.
    Distances equal the oracle's throughout."""
    from fuel_amd._lib import K_ESDF_ZY, K_ESDF_X
    map_size = (40.0, 20.0, 6.0)
    om = fo.OracleMap(map_size)
    gm = fa.SDFMap(map_size)
    nv = om.nvox
    assert nv == (400, 200, 60)
    occ = np.full(om.N, om.l_min).reshape(nv)       # known free ...
    occ[:, :, 0] = om.l_max                         # ... over an occupied floor
    occ[90:96, 100:104, 1:40] = om.l_max            # a pillar in the hall (x < 200)
    rng = np.random.default_rng(5)
    xs, ys, zs = np.meshgrid(np.arange(200, 400), np.arange(200), np.arange(60), indexing="ij")
    for _ in range(60):                             # the right half: blobs of unknown space
        c = rng.uniform([205, 5, 2], [395, 195, 55])
        r = rng.uniform(6, 14)
        occ[200:400][(xs - c[0]) ** 2 + (ys - c[1]) ** 2 + (zs - c[2]) ** 2 < r * r] = om.l_min - 0.01
    om.occ[:] = occ.reshape(-1)
    gm.uploadOccupancy(om.occ)
    hall = ((8, 8, 0), (191, 191, 59))
    fresh = ((208, 8, 0), (391, 191, 59))
    gm.profileEnable((1 << K_ESDF_ZY) | (1 << K_ESDF_X))
    ms = {"hall": [], "fresh": []}
    fams = {"hall": [], "fresh": []}
    for rnd in range(4):
        for name, (lo, hi) in (("hall", hall), ("fresh", fresh)):
            om.set_local_bound(lo, hi)
            gm.setLocalBound(lo, hi)
            om.inflate_local()
            om.update_esdf()
            gm.clearAndInflateLocalMap()
            gm.updateESDF3d()
            gm.synchronize()
            sl = tuple(slice(lo[k], hi[k] + 1) for k in range(3))
            h = gm.syncHost(distance=True, box=(lo, hi))
            assert np.abs(np.clip(h["distance"].reshape(nv)[sl], -BIG, BIG) -
                          np.clip(om.dist.reshape(nv)[sl], -BIG, BIG)).max() <= ESDF_TOL, (name, rnd)
            zy, xx = gm.profileSamples(K_ESDF_ZY), gm.profileSamples(K_ESDF_X)
            ms[name].append(zy[-1] + xx[-1])
            fams[name].append(gm.lastEsdfFamily())
    print("alternating local bound, ESDF ms per update: hall %s fresh %s" %
          (["%.3f" % v for v in ms["hall"]], ["%.3f" % v for v in ms["fresh"]]))
    # the hall gets the far-field kernels from its second visit on, the fresh region never loses the plain ones: the
    # statement under test is the kernel family each update ran (fuelmi_map_last_esdf_family), not a duration --
    # round 3 asserted event timings of 30-microsecond kernels here and failed one full-suite run in seven
    P, F = fa.SDFMap.ESDF_PLAIN, fa.SDFMap.ESDF_FAR
    assert fams["hall"] == [P, F, F, F], fams
    assert fams["fresh"] == [P, P, P, P], fams
    gm.close()


def test_optimiser_honours_the_wall_clock_cap(fa):
    """opt.set_maxtime(max_iteration_time_) (bspline_optimizer.cpp:170-172; 5 ms in algorithm.xml:190): a solve that
    runs out of time stops at the next evaluation boundary and returns the best variables seen so far, as
    costFunction keeps them (:693-707) -- fewer evaluations than the uncapped solve, a cost between the start's and the
    uncapped result's, and variables that reproduce the returned cost."""
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1])
    gm.uploadOccupancy(om.occ)
    lo, hi = helpers.full_box(om.nvox)
    om.set_local_bound(lo, hi)
    gm.setLocalBound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    cf = fa.NORMAL_PHASE | fa.MINTIME
    rng = np.random.default_rng(12)
    Cn, N, dt = 16, 32, 0.175
    ctrl = helpers.make_trajectories(rng, Cn, N, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
    x, ptd, st, en = helpers.bspline_inputs(ctrl, dt, True)
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    pb = fa.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 3, dt)
    dev = opt.deviceProblem(pb)
    c0, _ = opt.combineCost(pb)
    x_full, c_full, e_full = dev.optimize(max_eval=300)
    x_gen, c_gen, e_gen = dev.optimize(max_eval=300, max_time=1.0)     # a generous cap changes nothing
    assert np.array_equal(x_gen, x_full) and np.array_equal(e_gen, e_full)
    x_cap, c_cap, e_cap = dev.optimize(max_eval=300, max_time=200e-6)  # ~20 evaluations' worth of device time
    # (evaluation counts only: how MANY fewer depends on the box's clocks)
    assert (e_cap >= 1).all() and (e_cap <= e_full).all() and e_cap.sum() < e_full.sum(), (e_cap, e_full)
    assert (c_cap <= c0 + 1e-9).all() and (c_cap >= c_full - 1e-9).all()
    for c in range(Cn):
        chk, _ = fo.bspline_cost_grad(om, x_cap[c], N, cf, ptd[c], st[c], en[c], 3, 3, dt)
        assert abs(chk - c_cap[c]) <= 1e-6 * max(1.0, abs(chk))        # returned x <-> returned cost
    print("optimiser evaluations: uncapped %s; capped at 200 us %s" % (e_full.tolist(), e_cap.tolist()))
    gm.close()


def test_finder_that_outlives_its_map_refuses_politely(fa):
    """Garbage collectors and teardown orders destroy a map under a live finder: every entry point except _destroy
    must then return an error instead of dereferencing the map."""
    import ctypes as C
    from fuel_amd._lib import lib
    L = lib()
    gm = fa.SDFMap((6.0, 6.0, 3.0))
    gf = fa.FrontierFinder(gm, cluster_min=10)
    gm.close()                      # the map goes first
    n = C.c_int()
    assert L.fuelmi_frontier_search(gf.h, C.byref(n)) != 0
    assert L.fuelmi_frontier_search_begin(gf.h) != 0
    assert L.fuelmi_frontier_reset(gf.h) != 0
    assert L.fuelmi_frontier_commit(gf.h, 0) != 0
    assert L.fuelmi_frontier_synchronize(gf.h) != 0
    flags = np.empty(8, dtype=np.int8)
    assert L.fuelmi_frontier_get_flags(gf.h, flags.ctypes.data_as(C.c_char_p)) != 0
    assert b"destroyed" in L.fuelmi_last_error()
    gf.close()


def test_reference_order_many_clusters_and_a_large_one(fa):
    """reference_order on the two extremes of frontier_order.hip: hundreds of tiny clusters kept by
    cluster_min = 3 (legacy chain, two radix passes of the grouping: every cluster is swept inside LDS from the
    neighbour records of k_bfs_nbr) and, in the same search, the map-spanning cluster of the explored world
    (more cells than the LDS sweep holds when the map is large enough; here it exercises whichever class its
    size falls in).  Against the LITERAL oracle: cells in expandFrontier's order, sequential means
    (frontier_finder.cpp:123-164,374-390)."""
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    occ = om.occ.reshape(om.nvox)
    free = np.argwhere((occ >= om.l_min - 1e-3) & (occ <= om.l_occ))
    pick = free[np.all(free % 5 == 2, axis=1)]  # a lattice of isolated unknown voxels: separate little shells
    occ[tuple(pick.T)] = om.l_min - 0.01
    gm = fa.SDFMap((20.0, 20.0, 5.0), *box)
    gm.uploadOccupancy(om.occ)
    for cmin in (3, 0):
        of = fo.OracleFrontier(om, cmin)
        gf = fa.FrontierFinder(gm, cluster_min=cmin, reference_order=True)
        om.set_updated_box(*box)
        gm.setUpdatedBox(*box)
        n_o, n_g = of.search(), gf.searchFrontiers()
        assert n_o == n_g and n_o > 256
        ca, cb = of.clusters(0), gf.clusters(0)
        sizes = [len(c) for c in ca]
        assert max(sizes) > 20 * min(sizes)  # (tiny shells beside the explored world's own frontier)
        for k in range(n_o):
            assert np.array_equal(ca[k], cb[k]), "cluster %d (%d cells): order differs" % (k, sizes[k])
        for k in list(range(0, n_o, 41)) + [int(np.argmax(sizes))]:
            for x, y in zip(of.cluster_info(0, k), gf.clusterInfo(0, k)):
                assert np.array_equal(np.asarray(x), np.asarray(y)), "cluster %d: average_/box not bit-equal" % k
        assert np.array_equal(of.flags, gf.flags())
        gf.close()
    gm.close()


def test_reference_order_auto_reports_the_order_it_delivered(fa):
    """cfg.reference_order = 2 ("auto", the facade's default) answers per search -- and says which way
    (fuelmi_frontier_order_stats; VERDICT r3: the fallback used to be silent).  A search whose clusters all fit the
    in-LDS level sweep delivers the reference's BFS order, bit for bit; a search holding a cluster above the limit
    (26 624 cells) delivers ascending addresses, counts the fallback and names the cluster's size."""
    from fuel_amd._lib import lib
    import ctypes as C
    # (a) small clusters: the reference's order
    om, _, _, box = helpers.explored_oracle_map((9.0, 7.0, 4.0), 14, 25)
    gm = fa.SDFMap((9.0, 7.0, 4.0), *box)
    gm.uploadOccupancy(om.occ)
    of = fo.OracleFrontier(om, 10)
    gf = fa.FrontierFinder(gm, cluster_min=10, reference_order=2)
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    assert of.search() == gf.searchFrontiers() > 0
    for a, b in zip(of.clusters(0), gf.clusters(0)):
        assert np.array_equal(a, b), "auto mode on small clusters must deliver expandFrontier's order"
    assert gf.orderStats() == (1, 1, 0, 0)
    gf.close()
    gm.close()
    # (b) a map-spanning cluster: the address order, reported
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    gm = fa.SDFMap((20.0, 20.0, 5.0), *box)
    gm.uploadOccupancy(om.occ)
    of = fo.OracleFrontier(om, 100)
    gf = fa.FrontierFinder(gm, cluster_min=100, reference_order=2)
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    assert of.search() == gf.searchFrontiers() > 0
    ca, cb = of.clusters(0), gf.clusters(0)
    big = max(len(c) for c in ca)
    assert big > 26624, "the fixture is supposed to hold a cluster above the LDS sweep's limit (%d)" % big
    for a, b in zip(ca, cb):
        assert np.array_equal(np.sort(a), b), "the fallback is the ascending-address order of the same cells"
    last, n_ref, n_fallback, cells = gf.orderStats()
    assert (last, n_ref, n_fallback, cells) == (0, 0, 1, big)
    gf.close()
    # (c) the same search with reference_order = 1: the cluster above the LDS limit is swept through global memory
    # (k_bfs_sweep_g) and comes out in expandFrontier's order, sequential means included
    gf = fa.FrontierFinder(gm, cluster_min=100, reference_order=1)
    gm.setUpdatedBox(*box)
    assert gf.searchFrontiers() == len(ca)
    cb = gf.clusters(0)
    for k, (a, b) in enumerate(zip(ca, cb)):
        assert np.array_equal(a, b), "cluster %d (%d cells): order differs" % (k, len(a))
    kbig = int(np.argmax([len(c) for c in ca]))
    for x, y in zip(of.cluster_info(0, kbig), gf.clusterInfo(0, kbig)):
        assert np.array_equal(np.asarray(x), np.asarray(y)), "average_/box of the large cluster not bit-equal"
    assert gf.orderStats()[:2] == (1, 1)
    gf.close()
    # an invalid mode is refused, not treated as 0
    L = lib()
    cfg = fa._lib.FrontierCfg(100, 0.4, -1.0, -1, 0, 3)
    h = C.c_void_p()
    assert L.fuelmi_frontier_create(gm.h, C.byref(cfg), C.byref(h)) != 0
    gm.close()


def test_reference_order_of_a_sheet_whose_levels_outgrow_the_queue_ring(fa):
    """k_bfs_sweep_g keeps the BFS queue in an LDS ring and copies it out in bulk; a level too long to be re-read from
    the ring, or whose children might wrap onto entries not copied yet, goes through global memory instead (`direct`,
    `!in_ring` in frontier_order.hip).  The searches of the other tests never get there (their levels hold a few
    hundred cells).  A flat frontier sheet does: free space below an unknown ceiling over the whole 80 x 80 m box is
    ONE cluster of ~608 k cells whose BFS levels are square rings of up to ~1 560 cells (27 x 1 560 > 32 768 ring
    entries).  Cell order and sequential mean against the literal oracle (frontier_finder.cpp:123-164,374-390)."""
    map_size = (80.0, 80.0, 3.0)
    org = (-40.0, -40.0, -1.0)
    box = ((org[0] + 1.0, org[1] + 1.0, 0.0), (-org[0] - 1.0, -org[1] - 1.0, 1.4))
    om = fo.OracleMap(map_size, *box)
    nv = om.nvox
    assert nv == (800, 800, 30)
    occ = np.full(om.N, om.l_min).reshape(nv)   # known free ...
    occ[:, :, 15:] = om.l_min - 0.01            # ... below an unknown ceiling (world z >= 0.5)
    om.occ[:] = occ.reshape(-1)
    gm = fa.SDFMap(map_size, *box)
    gm.uploadOccupancy(om.occ)
    of = fo.OracleFrontier(om, 100)
    gf = fa.FrontierFinder(gm, cluster_min=100, reference_order=1)
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    n_o, n_g = of.search(), gf.searchFrontiers()
    assert n_o == n_g == 1
    a, b = of.clusters(0)[0], gf.clusters(0)[0]
    assert len(a) > 500000, len(a)
    assert np.array_equal(a, b), "first difference at position %d" % int(np.argmax(a != b))
    for x, y in zip(of.cluster_info(0, 0), gf.clusterInfo(0, 0)):
        assert np.array_equal(np.asarray(x), np.asarray(y)), "average_/box of the sheet not bit-equal"
    assert np.array_equal(of.flags, gf.flags())
    gf.close()
    gm.close()


@pytest.mark.parametrize("extra_row,cells", [(79, 26624), (80, 26625)])
def test_reference_order_at_the_lds_limit_exactly(fa, extra_row, cells):
    """The largest cluster the in-LDS level sweep takes (FR_REFORDER_AUTO = 26 624 cells: keys + queue fill the CU's
    160 KB) and the smallest one that goes through k_bfs_sweep_g (26 625), built to the cell: known free space under a
    140 x 152 patch of unknown ceiling plus `extra_row` columns of a 141st row -- floor cells and the patch's four
    walls are one frontier cluster.  Mode 1 delivers expandFrontier's order (frontier_finder.cpp:123-164) either way;
    mode 2 delivers it for the first and ascending addresses, reported, for the second."""
    map_size = (30.0, 30.0, 3.0)
    org = (-15.0, -15.0, -1.0)
    box = ((org[0] + 1.0, org[1] + 1.0, 0.0), (-org[0] - 1.0, -org[1] - 1.0, 1.4))
    om = fo.OracleMap(map_size, *box)
    nv = om.nvox
    occ = np.full(om.N, om.l_min).reshape(nv)
    occ[40:180, 40:192, 15:] = om.l_min - 0.01
    occ[180, 40:40 + extra_row, 15:] = om.l_min - 0.01
    om.occ[:] = occ.reshape(-1)
    gm = fa.SDFMap(map_size, *box)
    gm.uploadOccupancy(om.occ)
    of = fo.OracleFrontier(om, 100)
    om.set_updated_box(*box)
    assert of.search() == 1
    ref = of.clusters(0)[0]
    assert len(ref) == cells, "the fixture is built to hold exactly %d cells (%d)" % (cells, len(ref))
    for mode in (1, 2):
        gf = fa.FrontierFinder(gm, cluster_min=100, reference_order=mode)
        gm.setUpdatedBox(*box)
        assert gf.searchFrontiers() == 1
        got = gf.clusters(0)[0]
        in_lds = cells <= 26624
        if mode == 1 or in_lds:
            assert np.array_equal(ref, got), "mode %d: first difference at %d" % (mode, int(np.argmax(ref != got)))
            for x, y in zip(of.cluster_info(0, 0), gf.clusterInfo(0, 0)):
                assert np.array_equal(np.asarray(x), np.asarray(y))
            assert gf.orderStats() == (1, 1, 0, 0)
        else:
            assert np.array_equal(np.sort(ref), got)
            assert gf.orderStats() == (0, 0, 1, cells)
        gf.close()
    gm.close()
