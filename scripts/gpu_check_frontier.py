"""Ad-hoc GPU check: depth insert + frontier search vs oracle (development aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from oracle import fuel_oracle as fo
import fuel_amd

def cmp_clusters(oc, gc):
    oc = [np.sort(c) for c in oc]; gc = [np.sort(c) for c in gc]
    if len(oc) != len(gc): return False, "count %d vs %d" % (len(oc), len(gc))
    for k, (a, b) in enumerate(zip(oc, gc)):
        if not np.array_equal(a, b): return False, "cluster %d differs (%d vs %d cells)" % (k, len(a), len(b))
    return True, "%d clusters, %d cells" % (len(oc), sum(len(c) for c in oc))

def run(map_size, nobs, nframes, cluster_min, rounds=3, w=160, h=120):
    org = (-map_size[0] / 2.0, -map_size[1] / 2.0, -1.0)
    bmin = (org[0] + 1.0, org[1] + 1.0, 0.0); bmax = (-org[0] - 1.0, -org[1] - 1.0, max(0.8 * map_size[2] - 1.0, 1.0))
    om = fo.OracleMap(map_size, bmin, bmax)
    gm = fuel_amd.SDFMap(map_size, bmin, bmax)
    truth = om.fixture_world(42, nobs)
    of = fo.OracleFrontier(om, cluster_min); gf = fuel_amd.FrontierFinder(gm, cluster_min)
    k = 0
    for r in range(rounds):
        t_o = t_g = 0.0
        for _ in range(nframes):
            pose = om.fixture_camera(truth, 7, k, nframes * rounds, 0.7); k += 1
            pts = om.fixture_render(truth, pose, w, h, 2, 2)
            t = time.time(); om.input_points(pts, pose[:3]); t_o += time.time() - t
            t = time.time(); gm.inputPointCloud(pts, pose[:3]); t_g += time.time() - t
        h_ = gm.syncHost(occupancy=True)
        eq = np.array_equal(h_["occupancy"], om.occ)
        print(map_size, "round", r, "insert occ bit-equal:", eq, "unknown frac %.3f" % (om.occ < om.l_min - 1e-3).mean(),
              "lb", om.get_local_bound() == gm.getLocalBound(), "upd", np.array_equal(np.concatenate(om.get_updated_box()), np.concatenate(gm.getUpdatedBox())),
              "cpu %.3fs gpu %.3fs" % (t_o, t_g))
        if not eq:
            d = np.nonzero(h_["occupancy"] != om.occ)[0]; print("   ndiff", len(d), d[:5], h_["occupancy"][d[:5]], om.occ[d[:5]])
        t = time.time(); no = of.search(); t_o = time.time() - t
        t = time.time(); ng = gf.searchFrontiers(); t_g = time.time() - t
        ok, msg = cmp_clusters(of.clusters(0), gf.clusters(0))
        fl = np.array_equal(of.flags, gf.flags())
        print("   frontier new %d/%d clusters equal: %s (%s) flags equal: %s removed %s/%s cpu %.3fs gpu %.3fs" % (no, ng, ok, msg, fl, of.removed_ids(), gf.removedIds(), t_o, t_g))
        if no:
            i0 = of.cluster_info(0, 0); i1 = gf.clusterInfo(0, 0)
            print("   info diff", max(np.abs(a - b).max() for a, b in zip(i0, i1)))
        of.commit(r % 2 == 1); gf.commit(r % 2 == 1)
    gm.close()

run((8.0, 6.0, 4.0), 12, 8, 20)
run((20.0, 20.0, 5.0), 60, 15, 100, rounds=4)
run((40.0, 40.0, 10.0), 400, 20, 100, rounds=2, w=320, h=240)
