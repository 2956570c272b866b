"""Dev aid: depth-point fusion against the oracle with cameras at the map faces (the near-camera
LDS cube of k_insert_raycast is clipped by the map there), bit-exact log-odds after every frame."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import fuel_amd
from oracle import fuel_oracle as fo

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
import faulthandler; faulthandler.enable()
for size in [(6.0, 5.0, 3.0), (12.0, 3.2, 2.0), (3.0, 3.0, 6.4)]:
    om = fo.OracleMap(size)
    gm = fuel_amd.SDFMap(size)
    truth = om.fixture_world(9, 10)
    rng = np.random.default_rng(int(size[0] * 10))
    org = np.array([-size[0] / 2, -size[1] / 2, -1.0])
    for k in range(n_frames):
        # camera anywhere inside the map, often within a few voxels of a face (outside the map the reference
        # indexes its buffers out of bounds: undefined there, not a case to reproduce)
        u = rng.random(3)
        u = np.where(rng.random(3) < 0.4, np.round(u), u)  # snap some coordinates to a face
        cam = org + 0.06 + (np.array(size) - 0.12) * u
        n = int(rng.integers(50, 3000))
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1)[:, None]
        pts = (cam + d * (0.3 + 6.0 * rng.random((n, 1)))).astype(np.float32)
        om.input_points(pts, cam)
        gm.inputPointCloud(pts, cam)
        h = gm.syncHost(occupancy=True)["occupancy"]
        if not np.array_equal(h, om.occ):
            bad += 1
            print("MISMATCH size", size, "frame", k, int((h != om.occ).sum()), "voxels")
        lo_o, hi_o = om.get_local_bound()
        lo_g, hi_g = gm.getLocalBound()
        if tuple(lo_o) != tuple(lo_g) or tuple(hi_o) != tuple(hi_g):
            bad += 1
            print("BOUND MISMATCH", size, k, lo_o, lo_g, hi_o, hi_g)
    gm.close()
print("fuzz_insert done:", bad, "mismatches")
