"""ESDF updates only on a bench workload's map (for profilers / counters): esdf_only.py <workload> <family> [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, fuel_amd
from fuel_amd._lib import K_ESDF_ZY, K_ESDF_X
wl, fam = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
map_size, box, occ, ctrl, n_known = bench.build_inputs(wl, 42, 8)
fuel_amd.SDFMap.default_esdf_family = fam
kw = {"optimistic": 1} if wl in ("G400K", "G400E") else {}
m = fuel_amd.SDFMap(map_size, box[0], box[1], **kw)
m.uploadOccupancy(occ)
nv = m.nvox
m.setLocalBound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
m.clearAndInflateLocalMap()
m.profileEnable((1 << K_ESDF_ZY) | (1 << K_ESDF_X))
for _ in range(n):
    m.updateESDF3d()
m.synchronize()
print("zy ms median %.4f  x %.4f" % (np.median(m.profileSamples(K_ESDF_ZY)[1:]), np.median(m.profileSamples(K_ESDF_X)[1:])))
