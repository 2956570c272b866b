"""round 4: where a workgroup of the packed z/y pass spends its life (FUELMI_ZY_TIMING stamps), per tuning setting"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import bench, fuel_amd
from fuel_amd._lib import K_ESDF_ZY, K_ESDF_X
wl = sys.argv[1]
map_size, box, occ, ctrl, n_known = bench.build_inputs(wl, 42, 8)
fuel_amd.SDFMap.default_esdf_family = int(sys.argv[2])
m = fuel_amd.SDFMap(map_size, box[0], box[1])
m.uploadOccupancy(occ)
nv = m.nvox
m.setLocalBound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
m.clearAndInflateLocalMap()
m.profileEnable((1 << K_ESDF_ZY) | (1 << K_ESDF_X))
for _ in range(6):
    m.updateESDF3d()
m.synchronize()
print("zy ms median %%.4f  x %%.4f" %% (np.median(m.profileSamples(K_ESDF_ZY)[1:]), np.median(m.profileSamples(K_ESDF_X)[1:])))
''' % ROOT
def run(wl, fam, **env):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD, wl, str(fam)], env=e, capture_output=True, text=True, timeout=600)
    tim = [l for l in r.stderr.splitlines() if "zy-timing" in l]
    print(wl, "fam", fam, env, "|", r.stdout.strip(), "|", tim[-1] if tim else r.stderr[-300:])
    sys.stdout.flush()
for wl in sys.argv[1:] or ["G400"]:
    run(wl, 2)
    run(wl, 0, FUELMI_ZY_PP=0)
    run(wl, 0)
    run(wl, 0, FUELMI_ZY_TIMING=1)
    for sw in (2, 4, 6, 8, 12):
        run(wl, 0, FUELMI_ZY_PP_SCANW=sw)
    for wg in (1, 2, 3):
        run(wl, 0, FUELMI_ZY_PP_WGS=wg)
    run(wl, 0, FUELMI_ZY_PP_WGS=1, FUELMI_ZY_PP_SCANW=8)
    run(wl, 0, FUELMI_ZY_PP_WGS=1, FUELMI_ZY_PP_SCANW=12)
    run(wl, 0, FUELMI_ZY_PP_WGS=1, FUELMI_ZY_PP_SCANW=12, FUELMI_ZY_TIMING=1)
    run(wl, 0, FUELMI_ZY_FASTROW=0)
