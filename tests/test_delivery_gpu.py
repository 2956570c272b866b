"""Delivery of results to the host: the previous search's clusters stay readable across a reset
(list 3, fuelmi_frontier_keep_previous), voxel centres decoded by the library, B-spline results written by the kernel
into pinned slots -- each against what the plain calls return."""
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


def test_previous_search_stays_readable_and_centres_are_index_to_pos(fa):
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1])
    gm.uploadOccupancy(om.occ)
    gf = fa.FrontierFinder(gm, cluster_min=100)
    gf.keepPrevious(True)
    of = fo.OracleFrontier(om, 100)
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    assert of.search() == gf.searchFrontiers() > 0
    first = [np.sort(c) for c in of.clusters(0)]
    got = gf.clusters(0)
    assert all(np.array_equal(a, b) for a, b in zip(first, got))
    # centres = indexToPos of every cell (sdf_map.h:137-140)
    nv = np.array(om.nvox)
    for cells, cen in zip(got, gf.clusterCentres(0)):
        x = cells // (nv[1] * nv[2])
        r = cells - x * nv[1] * nv[2]
        idx = np.stack([x, r // nv[2], r % nv[2]], axis=1)
        want = (idx + 0.5) * om.res + np.array(om.origin)
        assert np.array_equal(cen, want)
    # a second, different search after a reset: the first one's clusters are list 3, bit for bit, while the new ones
    # are list 0 (the map changed in between: a slab of it becomes unknown)
    occ = om.occ.reshape(om.nvox).copy()
    occ[60:90, :, :] = om.l_min - 0.01
    om.occ[:] = occ.reshape(-1)
    gm.uploadOccupancy(om.occ)
    of2 = fo.OracleFrontier(om, 100)
    gf.reset()
    om.set_updated_box(*box)
    gm.setUpdatedBox(*box)
    assert of2.search() == gf.searchFrontiers() > 0
    prev = gf.clusters(3)
    assert len(prev) == len(first) and all(np.array_equal(a, b) for a, b in zip(first, prev))
    second = [np.sort(c) for c in of2.clusters(0)]
    assert all(np.array_equal(a, b) for a, b in zip(second, gf.clusters(0)))
    assert any(len(a) != len(b) for a, b in zip(first, second)) or len(first) != len(second)
    gf.reset()  # ... and the NEXT reset retires the second search instead
    prev2 = gf.clusters(3)
    assert len(prev2) == len(second) and all(np.array_equal(a, b) for a, b in zip(second, prev2))
    gf.keepPrevious(False)
    assert gf.clusters(3) == []
    gf.close()
    gm.close()


def test_pinned_bspline_slots_equal_the_plain_download(fa):
    om, _, _, box = helpers.explored_oracle_map((9.0, 7.0, 4.0), 14, 25)
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1])
    gm.uploadOccupancy(om.occ)
    lo, hi = helpers.full_box(om.nvox)
    gm.setLocalBound(lo, hi)
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    rng = np.random.default_rng(2)
    Cn, N, dt = 9, 20, 0.2
    ctrl = helpers.make_trajectories(rng, Cn, N, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5, seg_len=3.0)
    x, ptd, st, en = helpers.bspline_inputs(ctrl, dt, True)
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    pb = fa.BsplineBatchProblem(x, N, fa.NORMAL_PHASE | fa.MINTIME, ptd, st, en, 3, 3, dt)
    dev = opt.deviceProblem(pb)
    dev.eval()
    c0, g0 = dev.download()
    for slot in (0, 1, 0):
        dev.evalPinned(slot)
        c1, g1 = dev.collect(slot)
        assert np.array_equal(c0, c1) and np.array_equal(g0, g1)
    gm.close()
