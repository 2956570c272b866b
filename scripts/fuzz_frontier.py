"""Dev aid: frontier search (with and without splitting) against the oracle over many seeded maps and
incremental rounds -- hunts for order-dependent behaviour in the union-find / claim kernels."""
import sys, os
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import helpers
import fuel_amd
from oracle import fuel_oracle as fo

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for seed in range(n_seeds):
    size = [(12.0, 12.0, 4.0), (20.0, 16.0, 5.0), (16.0, 24.0, 3.0)][seed % 3]
    om, truth, frames, box = helpers.explored_oracle_map(size, 30 + 7 * (seed % 5), 6, seed=100 + seed, cam_seed=seed)
    gm = fuel_amd.SDFMap(tuple(om.cfg.map_size), box[0], box[1])
    for pts, cam in frames:  # same frames through the device fusion (keeps the updated boxes in step)
        gm.inputPointCloud(pts, cam)
    split = bool(seed & 1)
    of = fo.OracleFrontier(om, 20 + 10 * (seed % 4), cluster_size_xy=1.5, down_sample=3, split=split, canonical_order=True)
    gf = fuel_amd.FrontierFinder(gm, cluster_min=20 + 10 * (seed % 4), cluster_size_xy=1.5, down_sample=3, split=split)
    for rnd in range(4):
        if rnd > 0:  # new frames change the map: incremental search with remove_changed
            for k in range(3):
                pose = om.fixture_camera(truth, seed + 50 * rnd, k, 3, 0.7)
                pts = om.fixture_render(truth, pose, 160, 120, 2, 2)
                om.input_points(pts, pose[:3])
                gm.inputPointCloud(pts, pose[:3])
        n_o = of.search()
        n_g = gf.searchFrontiers()
        co = [np.sort(c) for c in of.clusters(0)]
        cg = [np.sort(c) for c in gf.clusters(0)]
        ok = n_o == n_g and len(co) == len(cg) and all(np.array_equal(a, b) for a, b in zip(co, cg))
        ok = ok and list(of.removed_ids()) == list(gf.removedIds())
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "round", rnd, n_o, n_g)
        of.commit(dormant=bool(rnd & 1))
        gf.commit(dormant=bool(rnd & 1))
    gm.close()
print("fuzz done:", n_seeds, "seeds,", bad, "mismatches")
