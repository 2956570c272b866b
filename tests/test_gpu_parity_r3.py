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
