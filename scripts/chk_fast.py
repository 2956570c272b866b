import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import helpers, fuel_amd as fa
from oracle import fuel_oracle as fo
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
        pose = om.fixture_camera(truth, 7, k, 60, 0.7); k += 1
        pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
        om.input_points(pts, pose[:3]); gm.inputPointCloud(pts, pose[:3])
    n_o, n_g = of.search(), gf.searchFrontiers()
    ok = n_o == n_g and all(np.array_equal(np.sort(a), b) for a, b in zip(of.clusters(0), gf.clusters(0)))
    print(r, n_o, n_g, ok, gf.stats(), np.array_equal(of.flags, gf.flags()))
    of.commit(); gf.commit()
