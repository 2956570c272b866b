"""Ad-hoc first GPU check of the core path against the oracle (development aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from oracle import fuel_oracle as fo
import fuel_amd

def run(map_size, nobs, nframes, boxes, optimistic=0, signed=0):
    om, truth, frames, (bmin, bmax) = helpers.explored_oracle_map(map_size, nobs, nframes, optimistic=optimistic, signed_dist=signed)
    gm = fuel_amd.SDFMap(map_size, bmin, bmax, optimistic=optimistic, signed_dist=signed)
    gm.uploadOccupancy(om.occ)
    for (lo, hi) in boxes:
        if lo is None: lo, hi = helpers.full_box(om.nvox)
        om.set_local_bound(lo, hi); gm.setLocalBound(lo, hi)
        t = time.time(); om.inflate_local(); om.update_esdf(); tc = time.time() - t
        gm.clearAndInflateLocalMap(); gm.updateESDF3d()
        h = gm.syncHost(occupancy=True, inflate=True, distance=True)
        print(map_size, "box", lo, hi, "cpu %.3fs" % tc,
              "occ eq", np.array_equal(h["occupancy"], om.occ),
              "infl eq", np.array_equal(h["inflate"], om.infl),
              "ninfl", int(om.infl.sum()))
        d_o = np.minimum(om.dist, 1e6); d_g = np.minimum(h["distance"], 1e6)
        print("   esdf max abs diff %.3e" % np.abs(d_o - d_g).max(), "max", d_o.max())
    # dist grad
    rng = np.random.default_rng(0)
    lo3 = om.origin - 0.3; hi3 = om.origin + np.array(map_size) + 0.3
    pos = lo3 + (hi3 - lo3) * rng.random((5000, 3))
    d0, g0 = om.dist_grad(pos); d1, g1 = gm.getDistWithGrad(pos)
    print("   distgrad diff", np.abs(np.minimum(d0,1e6) - np.minimum(d1,1e6)).max(), np.abs(np.clip(g0,-1e6,1e6) - np.clip(g1,-1e6,1e6)).max())
    # bspline
    Cn, N = 64, 32
    ctrl = helpers.make_trajectories(rng, Cn, N, np.array(bmin) + 0.5, np.array(bmax) - 0.5)
    for cf in (fuel_amd.NORMAL_PHASE | fuel_amd.MINTIME, fuel_amd.NORMAL_PHASE, fuel_amd.GUIDE_PHASE, fuel_amd.SMOOTHNESS | fuel_amd.WAYPOINTS, fuel_amd.NORMAL_PHASE | fuel_amd.VIEWCONS | fuel_amd.MINTIME):
        mint = bool(cf & fuel_amd.MINTIME)
        x, ptd, st, en = helpers.bspline_inputs(ctrl, 0.175, mint)
        guide = ctrl[:, 3:N-3, :] + 0.1
        wp = ctrl[:, [5, 11, 20], :] + 0.2; wi = np.tile(np.array([4, 10, 19], dtype=np.int32), (Cn, 1))
        vpt = ctrl[:, 10, :] + 0.5; vdir = np.tile(np.array([0.5, 1.0, 0.2]), (Cn, 1)); vidx = np.full(Cn, 12, dtype=np.int32)
        pb = fuel_amd.BsplineBatchProblem(x, N, cf, ptd, st, en, 3, 3, 0.175, 1.0 if mint else None, guide, wp, wi, vpt, vdir, vidx)
        opt = fuel_amd.BsplineOptimizer(ld_view=0.7); opt.setEnvironment(gm)
        cg, gg = opt.combineCost(pb)
        co = np.empty(Cn); go = np.empty_like(gg)
        for c in range(Cn):
            co[c], go[c] = fo.bspline_cost_grad(om, x[c], N, cf, ptd[c], st[c], en[c], 3, 3, 0.175, 1.0 if mint else -1.0, guide[c], wp[c], wi[c], (vpt[c], vdir[c], 12), ld_view=0.7)
        print("   bspline cf=%d cost rel %.2e grad abs %.2e (|g|max %.1f)" % (cf, np.abs(cg - co).max() / np.abs(co).max(), np.abs(gg - go).max(), np.abs(go).max()))
    gm.close()

run((8.0, 6.0, 4.0), 12, 20, [(None, None), ((10, 5, 3), (60, 40, 30))])
run((8.0, 6.0, 4.0), 12, 20, [(None, None)], optimistic=1, signed=1)
run((20.0, 20.0, 5.0), 60, 40, [(None, None), ((30, 50, 0), (150, 170, 49))])
