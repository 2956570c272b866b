"""Round-5 parity tests.  Regressions named by ADVICE r4 first: the legacy frontier chain across resets (each buffer
set has a per-search block of its own), centres of a cluster whose first cell lies in the voxel column x = y = 0."""
import numpy as np
import pytest

import helpers
from oracle import fuel_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import fuel_amd
    assert fuel_amd.lib().fuelmi_device_count() > 0, "no GPU visible: the HIP path cannot run"
    return fuel_amd


def sorted_clusters(cl):
    return [np.sort(c) for c in cl]


def test_legacy_chain_across_resets(fa):
    """cluster_min = 0 keeps a finder on the legacy chain (no tile path).  A reset swaps the finder's buffer sets, each
    with a device copy of the per-search block of its own: three fresh searches of three different maps, each equal
    to the oracle's (the second one read a block nothing had written)."""
    om, _, _, box = helpers.explored_oracle_map((12.0, 10.0, 4.0), 25, 20, width=120, height=90)
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1])
    gf = fa.FrontierFinder(gm, cluster_min=0)
    nv = om.nvox
    for rnd in range(3):
        if rnd:
            occ = om.occ.reshape(nv).copy()
            occ[20 * rnd:20 * rnd + 12, :, :] = om.unknown_value  # a slab goes back to unknown: other frontiers
            om.occ[:] = occ.reshape(-1)
        gm.uploadOccupancy(om.occ)
        of = fo.OracleFrontier(om, 0)
        gf.reset()
        om.set_updated_box(*box)
        gm.setUpdatedBox(*box)
        n_o, n_g = of.search(), gf.searchFrontiers()
        assert n_o == n_g > 0, "round %d" % rnd
        for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
            assert np.array_equal(a, b)
        assert np.array_equal(of.flags, gf.flags())
    assert gf.stats()[1] >= 3, gf.stats()  # (fast, legacy, fallbacks): every search ran the legacy chain
    gf.close()
    gm.close()


def test_centres_of_a_cluster_starting_in_the_corner_column(fa):
    """A frontier cluster whose first (lowest-address) cell is (0, 0, z) with z <= nz - 2: its centre is indexToPos of
    that cell like every other (the line-base start value made it (0, 0, z + 1))."""
    map_size = (4.0, 4.0, 2.0)
    org = np.array([-2.0, -2.0, -1.0])
    box = (tuple(org), tuple(org + np.array(map_size)))
    om = fo.OracleMap(map_size, *box)
    nv = om.nvox
    occ = np.full(nv, om.unknown_value)
    occ[:10, :10, :15] = om.l_min  # known free corner block; its top face z = 14 (0.45 m) is the frontier
    om.occ[:] = occ.reshape(-1)
    gm = fa.SDFMap(map_size, *box)
    gm.uploadOccupancy(om.occ)
    of = fo.OracleFrontier(om, 10)
    gf = fa.FrontierFinder(gm, cluster_min=10)
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    assert of.search() == gf.searchFrontiers() > 0
    got = gf.clusters(0)
    for a, b in zip(sorted_clusters(of.clusters(0)), got):
        assert np.array_equal(a, b)
    assert any(len(c) and c.min() <= nv[2] - 2 for c in got), "no cluster reaches the column x = y = 0"
    for cells, cen in zip(got, gf.clusterCentres(0)):
        x = cells // (nv[1] * nv[2])
        r = cells - x * nv[1] * nv[2]
        idx = np.stack([x, r // nv[2], r % nv[2]], axis=1)
        assert np.array_equal(cen, (idx + 0.5) * om.res + np.array(om.origin))
    gf.close()
    gm.close()


BIG = 1e6
ESDF_TOL = 1e-4  # north_star: ESDF values within 1e-4 m


def _esdf_equal(om, gm, lo, hi):
    h = gm.syncHost(distance=True)
    sl = tuple(slice(lo[i], hi[i] + 1) for i in range(3))
    d_o = np.clip(om.dist, -BIG, BIG).reshape(om.nvox)[sl]
    d_g = np.clip(h["distance"], -BIG, BIG).reshape(om.nvox)[sl]
    assert np.abs(d_o - d_g).max() <= ESDF_TOL, (lo, hi, float(np.abs(d_o - d_g).max()))


@pytest.mark.parametrize("map_size,optimistic,signed", [
    ((6.4, 4.8, 4.0), 0, 0),   # nz = 40: z range in pieces of 8 + 2 segments
    ((5.0, 4.4, 2.8), 1, 0),   # nz = 28: 4 + 2 + 1 segments (no full-width tile at all)
    ((5.0, 4.4, 2.8), 0, 1),   # ... signed: the negative field through the same 16-bit hand-over, merged in place
    ((4.2, 3.8, 10.0), 1, 1),  # nz = 100: 8 + 8 + 8 + 1 (the headline map's z extent)
])
def test_packed_esdf_hand_over_over_box_shapes(fa, map_size, optimistic, signed):
    """The packed family's 16-bit, tile-contiguous hand-over (esdf.hip k_esdf_zy_pk2 / k_esdf_x_pk2, round 5) on boxes
    that exercise its geometry: odd and even x / y extents (the last slab pair and the last row pair repeat their
    partner), boxes starting at odd x, z ranges that are not 4-aligned at either end, y extents that do not fill the
    remainder tiles, single-voxel and single-slab boxes.  Family pinned to the packed one and asserted to have run."""
    om, _, _, box = helpers.explored_oracle_map(map_size, 10, 14, width=120, height=90, optimistic=optimistic,
                                                signed_dist=signed)
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1], optimistic=optimistic, signed_dist=signed)
    gm.uploadOccupancy(om.occ)
    gm.setEsdfFamily(0)
    nv = om.nvox
    rng = np.random.default_rng(nv[2])
    boxes = [helpers.full_box(nv), ((1, 0, 0), (nv[0] - 1, nv[1] - 2, nv[2] - 1)), ((3, 2, 1), (nv[0] - 2, nv[1] - 1, nv[2] - 2)),
             ((5, 5, 2), (5, 20, 13)), ((0, 7, 5), (30, 7, 6)), ((9, 9, 9), (9, 9, 9)), ((2, 1, 3), (17, 12, 3))]
    for _ in range(6):
        a = [int(rng.integers(0, nv[k] - 1)) for k in range(3)]
        b = [int(rng.integers(a[k], nv[k])) for k in range(3)]
        boxes.append((tuple(a), tuple(b)))
    for lo, hi in boxes:
        om.set_local_bound(lo, hi)
        gm.setLocalBound(lo, hi)
        om.inflate_local()
        om.update_esdf()
        gm.clearAndInflateLocalMap()
        gm.updateESDF3d()
        assert gm.lastEsdfFamily() == 0, "the packed family did not run on box %s %s" % (lo, hi)
        _esdf_equal(om, gm, lo, hi)
    gm.close()


@pytest.mark.parametrize("map_size,src", [((32.0, 1.2, 0.8), (4, 5, 3)), ((1.2, 32.0, 0.8), (5, 4, 3)), ((2.0, 30.0, 0.8), (17, 290, 2))])
def test_packed_esdf_far_from_every_source_uses_the_wide_plane(fa, map_size, src):
    """Outputs further than 255 voxels from every source leave the 16-bit range.  Along x: the x pass recomputes them from
    its tile.  Along y: the z/y pass marks them in the hand-over and keeps the exact value in the wide plane, the x pass
    reads it back.  One occupied voxel in a 300-voxel-long known corridor, optimistic map (only the inflated voxel block
    is a source)."""
    om = fo.OracleMap(map_size, optimistic=1)
    gm = fa.SDFMap(map_size, optimistic=1)
    nv = om.nvox
    occ = np.full(nv, om.l_min)
    occ[src] = om.l_max
    om.occ[:] = occ.reshape(-1)
    gm.uploadOccupancy(om.occ)
    gm.setEsdfFamily(0)
    lo, hi = helpers.full_box(nv)
    for o in (om,):
        o.set_local_bound(lo, hi)
        o.inflate_local()
        o.update_esdf()
    gm.setLocalBound(lo, hi)
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    assert gm.lastEsdfFamily() == 0
    assert np.nanmax(np.where(om.dist < BIG, om.dist, 0.0)) > 25.6  # (there ARE distances beyond 255 voxels)
    _esdf_equal(om, gm, lo, hi)
    gm.close()


def test_fusion_beside_a_running_search_waits_for_its_plane_reads(fa):
    """A finder marks the point behind which the occupancy planes may be rewritten (an event between the chain's first two
    kernels) only once it has seen a mutator arrive beside one of its searches; the FIRST such mutator finds no mark and
    waits for what is queued (fuelmi_map::late_readers), the later ones for the mark.  Either way the running search must
    see the planes of ITS frame: every search below equals the oracle's on the state before the frame fused beside it."""
    map_size = (12.0, 10.0, 4.0)
    box = ((-5.0, -4.0, 0.0), (5.0, 4.0, 2.4))
    om = fo.OracleMap(map_size, *box)
    gm = fa.SDFMap(map_size, *box)
    truth = om.fixture_world(11, 20)
    of = fo.OracleFrontier(om, 20)
    gf = fa.FrontierFinder(gm, cluster_min=20)
    n = 7
    frames = []
    for k in range(n):
        pose = om.fixture_camera(truth, 3, k, n, 0.7)
        frames.append((om.fixture_render(truth, pose, 160, 120, 2, 2), pose[:3].copy()))
    om.input_points(*frames[0])
    gm.inputPointCloud(*frames[0])
    for k in range(n):
        gf.searchFrontiersBegin()            # search of frame k ...
        if k + 1 < n:
            gm.inputPointCloud(*frames[k + 1])  # ... and frame k + 1 fused beside it (k = 0: the finder has no mark yet)
        n_o = of.search()                       # the oracle: frame k's state only
        n_g = gf.searchFrontiersEnd()
        assert n_o == n_g, "frame %d: %d vs %d new clusters" % (k, n_g, n_o)
        for a, b in zip(sorted_clusters(of.clusters(0)), gf.clusters(0)):
            assert np.array_equal(a, b), "frame %d" % k
        assert np.array_equal(of.removed_ids(), gf.removedIds())
        of.commit()
        gf.commit()
        if k + 1 < n:
            om.input_points(*frames[k + 1])
    assert np.array_equal(of.flags, gf.flags())
    h = gm.syncHost(occupancy=True)
    assert np.array_equal(h["occupancy"], om.occ)
    gf.close()
    gm.close()


def test_full_size_g800_esdf_properties(fa):
    """BASELINE's largest map (800 x 800 x 200, bench.py --workload G800: the HBM-resident roofline run), full box, packed
    kernels with the 16-bit hand-over: too large for the oracle in a test, so size-independent properties of an exact
    Euclidean distance transform instead -- (a) bit-identical to the 32-bit kernel family (both claim exact integer squared
    distances: any disagreement is a bug in one of them), (b) zero exactly on the sources, positive elsewhere, (c) squared
    voxel distances are integers, (d) 1-Lipschitz along all three axes (|d(p) - d(q)| <= res for face neighbours),
    (e) idempotent: a second update rewrites the same bits."""
    import bench
    map_size, box, occ, _, _ = bench.build_inputs("G800", seed=42, n_traj=1)
    gm = fa.SDFMap(map_size, box[0], box[1])
    gm.uploadOccupancy(occ)
    nv = gm.nvox
    assert nv == (800, 800, 200)
    lo, hi = helpers.full_box(nv)
    gm.setLocalBound(lo, hi)
    gm.clearAndInflateLocalMap()
    gm.setEsdfFamily(0)
    gm.updateESDF3d()
    assert gm.lastEsdfFamily() == 0
    h = gm.syncHost(inflate=True, distance=True)
    d0 = h["distance"].reshape(nv).astype(np.float32)
    infl = h["inflate"].reshape(nv)
    gm.updateESDF3d()                                   # (e)
    assert np.array_equal(gm.syncHost(distance=True)["distance"].reshape(nv).astype(np.float32), d0)
    gm.setEsdfFamily(2)                                 # (a)
    gm.updateESDF3d()
    assert gm.lastEsdfFamily() == 2
    d2 = gm.syncHost(distance=True)["distance"].reshape(nv).astype(np.float32)
    assert np.array_equal(d2, d0), "packed and 32-bit families disagree on %d voxels" % int((d2 != d0).sum())
    del d2
    o3 = np.asarray(occ).reshape(nv)
    info = gm.info
    unknown = o3 < info.clamp_min_log - 1e-3
    src = (infl == 1) | unknown                          # (b) sources of a non-optimistic map: inflated or unknown
    assert np.all(d0[src] == 0.0) and np.all(d0[~src] > 0.0)
    res = np.float32(0.1)
    for sl in (np.s_[::97, :, :], np.s_[:, ::89, :], np.s_[:, :, ::23]):   # (c) on a sample of slabs
        sq = (d0[sl].astype(np.float64) / 0.1) ** 2
        fin = np.isfinite(sq) & (sq < 1e9)
        assert np.abs(sq[fin] - np.rint(sq[fin])).max() < 5e-2
    for ax in range(3):                                  # (d)
        a = np.take(d0, range(0, nv[ax] - 1), axis=ax)
        b = np.take(d0, range(1, nv[ax]), axis=ax)
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.abs(a[fin] - b[fin]).max() <= res * (1 + 1e-5)
    gm.close()
