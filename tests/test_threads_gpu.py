"""One map, many threads (SURVEY 8b "queries must be re-entrant").

The reference calls the hot path from one ros::spin thread, but READS one map from several: up to ten optimiser
threads in topoReplan (plan_manage/src/planner_manager.cpp:446-453) and the detached visualisation thread
(exploration_manager/src/fast_exploration_fsm.cpp:122).  Here ten threads hammer ONE map with getDistWithGrad,
combineCost and whole optimize() solves while the owner thread keeps running inflate -> ESDF -> frontier search on it;
every result must equal the serial run's bit for bit (the queries run on the map's query slots: side streams, pinned
blocks, no allocation; the mutators rewrite the same values)."""
import threading

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


@pytest.mark.parametrize("signed", [0, 1])
def test_ten_reader_threads_beside_the_mutating_owner(fa, signed):
    """signed = 1: with signed_dist the x pass of the negative field merges into the distance buffer in
    place -- a reader overlapping the update must never see the positive-only intermediate; every reader result below is
    compared bit for bit with the serial run."""
    om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40, signed_dist=signed)
    gm = fa.SDFMap(tuple(om.cfg.map_size), box[0], box[1], signed_dist=signed)
    gm.uploadOccupancy(om.occ)
    lo, hi = helpers.full_box(om.nvox)
    gm.setLocalBound(lo, hi)
    gm.clearAndInflateLocalMap()
    gm.updateESDF3d()
    gm.synchronize()
    gf = fa.FrontierFinder(gm, cluster_min=100)
    rng = np.random.default_rng(3)
    n_thr = 10
    cf = fa.NORMAL_PHASE | fa.MINTIME
    N, dt = 24, 0.175
    work = []
    opt = fa.BsplineOptimizer()
    opt.setEnvironment(gm)
    for t in range(n_thr):
        pos = np.array(box[0]) + (np.array(box[1]) - np.array(box[0])) * rng.random((700 + 37 * t, 3))
        ctrl = helpers.make_trajectories(rng, 3, N, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
        x, ptd, st, en = helpers.bspline_inputs(ctrl, dt, True)
        pb = fa.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 3, dt)
        d, g = gm.getDistWithGrad(pos)
        c0, g0 = opt.combineCost(pb)
        xs, cs, es = opt.optimize(pb, max_eval=60)
        work.append((pos, pb, d, g, c0, g0, xs, cs, es))
    # spot check of the serial results against the oracle (the parity suite does this at length)
    om.set_local_bound(lo, hi)
    om.inflate_local()
    om.update_esdf()
    chk, _ = fo.bspline_cost_grad(om, work[0][1].x[0], N, cf, work[0][1].pt_dist[0], work[0][1].start_state[0], work[0][1].end_state[0], 3, 3, dt)
    assert abs(chk - work[0][4][0]) <= 1e-6 * max(1.0, abs(chk))
    stop = threading.Event()
    errs = []

    def reader(t):
        try:
            o = fa.BsplineOptimizer()
            o.setEnvironment(gm)
            pos, pb, d, g, c0, g0, xs, cs, es = work[t]
            for it in range(12):
                d2, g2 = gm.getDistWithGrad(pos)
                assert np.array_equal(d2, d) and np.array_equal(g2, g), "getDistWithGrad differs (thread %d, pass %d)" % (t, it)
                c2, gg2 = o.combineCost(pb)
                assert np.array_equal(c2, c0) and np.array_equal(gg2, g0), "combineCost differs (thread %d)" % t
                if it % 3 == 0:
                    x2, cc2, e2 = o.optimize(pb, max_eval=60)
                    assert np.array_equal(x2, xs) and np.array_equal(cc2, cs) and np.array_equal(e2, es), \
                        "optimize() differs (thread %d)" % t
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    def owner():
        try:
            n = 0
            while not stop.is_set() or n < 5:
                gf.reset()
                gm.setUpdatedBox(box[0], box[1])
                gf.searchFrontiersBegin()
                gm.clearAndInflateLocalMap()
                gm.updateESDF3d()
                gf.searchFrontiersEnd()
                n += 1
            gm.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append("owner: " + repr(e))

    th = [threading.Thread(target=reader, args=(t,)) for t in range(n_thr)]
    ow = threading.Thread(target=owner)
    ow.start()
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    stop.set()
    ow.join(timeout=600)
    assert not errs, errs[:3]
    # ... and the map the owner kept rewriting still answers like the serial run
    d2, g2 = gm.getDistWithGrad(work[0][0])
    assert np.array_equal(d2, work[0][2]) and np.array_equal(g2, work[0][3])
    gf.close()
    gm.close()
