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
