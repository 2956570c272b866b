"""Tuning aid: time of one batched B-spline cost/gradient launch per subset of cost terms."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench, fuel_amd

map_size, box, occ, ctrl, _ = bench.build_inputs("G100", 42, 64)
gm = fuel_amd.SDFMap(map_size, box[0], box[1], device=0)
gm.uploadOccupancy(occ)
nv = gm.nvox
gm.setLocalBound((0, 0, 0), (nv[0] - 1, nv[1] - 1, nv[2] - 1))
gm.clearAndInflateLocalMap()
gm.updateESDF3d()
x, ptd, st, en = bench.bspline_problem(ctrl, 0.175)
opt = fuel_amd.BsplineOptimizer()
opt.setEnvironment(gm)
F = fuel_amd
sets = {"all (NORMAL|MINTIME)": F.NORMAL_PHASE | F.MINTIME, "smooth": F.SMOOTHNESS, "dist": F.DISTANCE,
        "feasi": F.FEASIBILITY, "start": F.START, "end": F.END, "start+end": F.START | F.END,
        "smooth+dist+feasi": F.SMOOTHNESS | F.DISTANCE | F.FEASIBILITY, "mintime only": F.MINTIME}
for name, cf in sets.items():
    xx = x if cf & F.MINTIME else x[:, :-1].copy()
    dev = opt.deviceProblem(F.BsplineBatchProblem(xx, ctrl.shape[1], cf, ptd, st, en, 3, 3, 0.175))
    for _ in range(20):
        dev.eval()
    gm.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        dev.eval()
    gm.synchronize()
    print("%-24s %.2f us per launch" % (name, (time.perf_counter() - t0) / 400 * 1e6))
