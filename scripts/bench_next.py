"""Measurement of the SURVEY 8(f) 'next' rows on the bench map (G400): GPU (C-ABI) beside the CPU oracle.
Prints one JSON object; copy it to profiles/ (dev aid, not part of bench.py's contract)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fuel_amd
from fuel_amd import synth
from oracle import fuel_oracle as fo

out = {}
map_size, box, occ, ctrl, n_known = bench.build_inputs("G400", seed=42)

# ---- rank 2: searchFrontiers incl. splitLargeFrontiers -------------------------------------------
gm = fuel_amd.SDFMap(map_size, box[0], box[1], device=0)
gm.uploadOccupancy(occ)
gf = fuel_amd.FrontierFinder(gm, cluster_min=100, cluster_size_xy=2.0, down_sample=3, split=True)
ts = []
for it in range(12):
    gf.reset()
    gm.setUpdatedBox(box[0], box[1])
    t0 = time.perf_counter()
    n = gf.searchFrontiers()
    ts.append(time.perf_counter() - t0)
gpu_ms = float(np.median(ts[2:]) * 1e3)
om = fo.OracleMap(map_size, box[0], box[1])
om.occ[:] = occ
of = fo.OracleFrontier(om, cluster_min=100, cluster_size_xy=2.0, down_sample=3, split=True, canonical_order=True)
om.set_updated_box(box[0], box[1])
t0 = time.perf_counter()
n_o = of.search()
cpu_ms = (time.perf_counter() - t0) * 1e3
out["frontier_search_with_split"] = {"gpu_ms": gpu_ms, "cpu_oracle_ms": cpu_ms, "pieces_gpu": n, "pieces_cpu": n_o,
                                     "cells": int(sum(len(c) for c in gf.clusters(0)))}

# the same search in reference_order (cells in expandFrontier's BFS order, sequential means, in-order VoxelGrid
# sums: bit-exact against the reference) -- what the mode costs; checked against the literal oracle
gfr = fuel_amd.FrontierFinder(gm, cluster_min=100, cluster_size_xy=2.0, down_sample=3, split=True, reference_order=True)
tr = []
for it in range(6):
    gfr.reset()
    gm.setUpdatedBox(box[0], box[1])
    t0 = time.perf_counter()
    nr = gfr.searchFrontiers()
    tr.append(time.perf_counter() - t0)
ofl = fo.OracleFrontier(om, cluster_min=100, cluster_size_xy=2.0, down_sample=3, split=True)
om.set_updated_box(box[0], box[1])
t0 = time.perf_counter()
n_l = ofl.search()
cpu_l_ms = (time.perf_counter() - t0) * 1e3
same = n_l == nr and all(np.array_equal(a, b) for a, b in zip(ofl.clusters(0), gfr.clusters(0))) and \
    all(np.array_equal(ofl.cluster_info(0, k)[0], gfr.clusterInfo(0, k)[0]) for k in range(nr))
gfu = fuel_amd.FrontierFinder(gm, cluster_min=100, reference_order=True)  # without splitting
tu = []
for it in range(6):
    gfu.reset()
    gm.setUpdatedBox(box[0], box[1])
    t0 = time.perf_counter()
    nu = gfu.searchFrontiers()
    tu.append(time.perf_counter() - t0)
out["frontier_search_reference_order"] = {"gpu_ms_with_split": float(np.median(tr[1:]) * 1e3),
                                          "gpu_ms_without_split": float(np.median(tu[1:]) * 1e3),
                                          "gpu_ms_default_order_with_split": gpu_ms, "cpu_oracle_literal_ms": cpu_l_ms,
                                          "pieces": int(nr), "clusters_unsplit": int(nu),
                                          "bit_exact_vs_literal_oracle": bool(same)}
gfr.close()
gfu.close()

# ---- rank 1: viewpoint sampling for every piece (computeFrontiersToVisit) -----------------------------
gm.setLocalBound((0, 0, 0), tuple(v - 1 for v in gm.nvox))
gm.clearAndInflateLocalMap()
gf.setViewpointConfig(gf.viewpointConfig())
gm.synchronize()
t0 = time.perf_counter()
na, nd = gf.computeFrontiersToVisit()
gpu_vp_ms = (time.perf_counter() - t0) * 1e3
nvp = sum(len(gf.viewpoints(1, k)[1]) for k in range(na))
om.set_local_bound((0, 0, 0), tuple(v - 1 for v in om.nvox))
om.inflate_local()
of.set_viewpoint_cfg(fo.viewpoint_cfg())
t0 = time.perf_counter()
of.compute_to_visit()
cpu_vp_ms = (time.perf_counter() - t0) * 1e3
out["compute_frontiers_to_visit"] = {"gpu_ms": gpu_vp_ms, "cpu_oracle_ms": cpu_vp_ms, "active": na, "dormant": nd,
                                     "active_cpu": len(of.clusters(1)), "viewpoints": int(nvp),
                                     "candidates": int(n * 100)}

# ---- rank 4: whole trajectory solves (BsplineOptimizer::optimize) for the 64 bench candidates ----------
gm.updateESDF3d()
om.update_esdf()
x, ptd, st, en = bench.bspline_problem(ctrl, 0.175)
cf = fuel_amd.NORMAL_PHASE | fuel_amd.MINTIME
opt = fuel_amd.BsplineOptimizer()
opt.setEnvironment(gm)
pb = fuel_amd.BsplineBatchProblem(x, ctrl.shape[1], cf, ptd, st, en, 3, 3, 0.175)
dev = opt.deviceProblem(pb)
dev.optimize(max_eval=300)
gm.synchronize()
t0 = time.perf_counter()
xg, cg, eg = dev.optimize(max_eval=300)
gpu_opt_ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
co, eo = [], []
for c in range(8):  # bounded CPU sample: 8 of the 64 solves
    xo, f, e = fo.bspline_optimize(om, x[c], ctrl.shape[1], cf, ptd[c], st[c], en[c], 3, 3, 0.175, max_eval=300)
    co.append(f)
    eo.append(e)
cpu_opt_ms = (time.perf_counter() - t0) * 1e3 * (len(x) / 8.0)
# the solve is latency-bound per candidate (one wavefront each): a larger batch costs almost nothing more
rng = np.random.default_rng(5)
ctrl_big = np.concatenate([ctrl + rng.normal(scale=0.05, size=ctrl.shape) for _ in range(16)], axis=0)
xb, ptdb, stb, enb = bench.bspline_problem(ctrl_big, 0.175)
devb = opt.deviceProblem(fuel_amd.BsplineBatchProblem(xb, ctrl.shape[1], cf, ptdb, stb, enb, 3, 3, 0.175))
devb.optimize(max_eval=300)
t0 = time.perf_counter()
_, _, egb = devb.optimize(max_eval=300)
gpu_big_ms = (time.perf_counter() - t0) * 1e3
out["bspline_optimize_1024_candidates"] = {"gpu_ms": gpu_big_ms, "evals_gpu_mean": float(egb.mean()),
                                            "cpu_oracle_ms_extrapolated": cpu_opt_ms * 16}
out["bspline_optimize_64_candidates"] = {"gpu_ms": gpu_opt_ms, "cpu_oracle_ms_extrapolated_from_8": cpu_opt_ms,
                                          "evals_gpu_mean": float(eg.mean()), "evals_cpu_mean": float(np.mean(eo)),
                                          "final_cost_ratio_gpu_over_cpu_first8": float(np.mean(cg[:8] / np.array(co)))}

# ---- rank 4 glue: samples -> parameterizeToBspline -> getBoundaryStates -> optimize, 1024 candidates -------
K = ctrl.shape[1] - 2
smp = np.stack([(c[:-2] + 4 * c[1:-1] + c[2:]) / 6.0 for c in ctrl_big])  # the splines' own knot points
der = np.zeros((len(smp), 4, 3))
tsb = np.full(len(smp), 0.175)
x0 = np.zeros((len(smp), 3 * ctrl.shape[1] + 1))
x0[:, -1] = 1.0
devs = opt.deviceProblem(fuel_amd.BsplineBatchProblem(x0, ctrl.shape[1], cf, np.ones(len(smp)),
                                                      np.zeros((len(smp), 3, 3)), np.zeros((len(smp), 3, 3)), 1, 3, 1.0))
devs.loadSamples(tsb, smp, der)
devs.optimize(max_eval=300)
t0 = time.perf_counter()
devs.loadSamples(tsb, smp, der)
gm.synchronize()
gpu_fit_ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
devs.loadSamples(tsb, smp, der)
devs.optimize(max_eval=300)
gpu_fit_solve_ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
for c in range(64):  # bounded CPU sample
    cp = fo.spline_parameterize(tsb[c], smp[c], der[c], 3)
    fo.spline_boundary_states(cp, tsb[c], 3, 2, 0)
    fo.bspline_pt_dist(cp)
cpu_fit_ms = (time.perf_counter() - t0) * 1e3 * (len(smp) / 64.0)
out["samples_to_control_points_1024_candidates"] = {"gpu_ms": gpu_fit_ms, "cpu_oracle_ms_extrapolated_from_64": cpu_fit_ms,
                                                     "samples_per_candidate": int(K)}
out["samples_to_solved_trajectory_1024_candidates"] = {"gpu_ms": gpu_fit_solve_ms}

# ---- rank 3: depth frame -> fused map ----------------------------------------------------------------
w = synth.World.for_map_size(map_size)
truth = w.world(42, bench.WORKLOADS["G400"][1])
gm2 = fuel_amd.SDFMap(map_size, box[0], box[1], device=0)
om2 = fo.OracleMap(map_size, box[0], box[1])
frames = []
for k in range(12):
    pose = w.camera(truth, 7, k, 12, 0.7)
    frames.append((w.depth_image(truth, pose, 640, 480), pose, synth.World.pose_quaternion(pose)))
tg, tc = [], []
for img, pose, q in frames:
    t0 = time.perf_counter()
    npts = gm2.inputDepthImage(img, pose[:3], q)
    gm2.synchronize()
    tg.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    pts = fo.project_depth(img, pose[:3], q)
    om2.input_points(pts, pose[:3])
    tc.append(time.perf_counter() - t0)
out["depth_frame_640x480_project_and_fuse"] = {"gpu_ms": float(np.median(tg[2:]) * 1e3),
                                               "cpu_oracle_ms": float(np.median(tc[2:]) * 1e3), "points": int(npts)}
# ---- the complete exploration front end per planning cycle, as the facade drives it -----------------------
# depth frame -> fusion -> inflation -> ESDF -> search incl. splitting -> viewpoints -> 64 full solves
gm3 = fuel_amd.SDFMap(map_size, box[0], box[1], device=0)
om3 = fo.OracleMap(map_size, box[0], box[1])
gf3 = fuel_amd.FrontierFinder(gm3, cluster_min=100, cluster_size_xy=2.0, down_sample=3, split=True)
gf3.setViewpointConfig(gf3.viewpointConfig())
of3 = fo.OracleFrontier(om3, cluster_min=100, cluster_size_xy=2.0, down_sample=3, split=True, canonical_order=True)
of3.set_viewpoint_cfg(fo.viewpoint_cfg())
opt3 = fuel_amd.BsplineOptimizer()
opt3.setEnvironment(gm3)
dev3 = opt3.deviceProblem(fuel_amd.BsplineBatchProblem(x, ctrl.shape[1], cf, ptd, st, en, 3, 3, 0.175))
tg3, tc3, parts3 = [], [], []
for img, pose, q in frames:
    t0 = time.perf_counter()
    if gm3.inputDepthImage(img, pose[:3], q) > 0:
        gm3.clearAndInflateLocalMap()
        gm3.updateESDF3d()
    t1 = time.perf_counter()
    gf3.searchFrontiers()
    t2 = time.perf_counter()
    gf3.computeFrontiersToVisit()
    t3 = time.perf_counter()
    dev3.optimize(max_eval=100)
    gm3.synchronize()
    t4 = time.perf_counter()
    tg3.append(t4 - t0)
    parts3.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    t0 = time.perf_counter()
    pts = fo.project_depth(img, pose[:3], q)
    if len(pts):
        om3.input_points(pts, pose[:3])
        om3.inflate_local()
        om3.update_esdf()
    of3.search()
    of3.compute_to_visit()
    for c in range(4):  # bounded sample of the 64 solves
        fo.bspline_optimize(om3, x[c], ctrl.shape[1], cf, ptd[c], st[c], en[c], 3, 3, 0.175, max_eval=100)
    tc3.append(time.perf_counter() - t0)
out["full_front_end_cycle"] = {"gpu_ms": float(np.median(tg3[2:]) * 1e3),
                               "cpu_oracle_ms_with_4_of_64_solves": float(np.median(tc3[2:]) * 1e3),
                               "gpu_ms_map_search_viewpoints_solves": [round(float(v) * 1e3, 3) for v in np.median(np.array(parts3[2:]), axis=0)],
                               "active_frontiers_gpu": len(gf3.clusters(1)), "active_frontiers_cpu": len(of3.clusters(1))}

# ---- what the reference's cell order costs on the incremental searches of the streaming workload (config #4) ----
# the same 100 frames with frontier reference_order 0 (address order), 1 (always the BFS order) and 2 (BFS order
# while every cluster fits the LDS sweep: the facade's default), each on a fresh map, issued by the C++ loop
row = {}
# (the maps and finders of the rows above are closed first: HIP multiplexes a process's streams onto a handful of
# hardware queues, and with enough of them alive a map's stream and its finder's can land on the same queue -- the
# frame then serialises, 0.11 -> 0.37 ms, scripts/r4_hwq_probe.py; this row is about the cell order, not about that)
for _o in ("dev3", "gf3", "gm3", "gm2", "devs", "devb", "dev", "gf", "gm"):
    try:
        globals()[_o].close()
    except Exception:
        pass
try:
    map_size_s, n_obs_s, _ = bench.WORKLOADS["G800S"]
    box_s = bench.exploration_box(map_size_s)
    frames_s = bench.streaming_frames(map_size_s, n_obs_s, 124, seed=42)
    ctrl_s = bench.make_trajectories(np.random.default_rng(1042), 64, 32, np.array(box_s[0]) + 0.5, np.array(box_s[1]) - 0.5)
    for ro in (0, 1, 2):
        cyc = bench.GpuStreamCycle(map_size_s, box_s, frames_s, ctrl_s, device=0, reference_order=ro)
        cyc.run_native(20)
        cyc.finish()
        sec = cyc.run_native(100)
        cyc.finish()
        row["reference_order_%d_ms_per_frame" % ro] = round(sec / 100 * 1e3, 4)
        cyc.close()
    row["added_by_order_1_ms"] = round(row["reference_order_1_ms_per_frame"] - row["reference_order_0_ms_per_frame"], 4)
    out["streaming_reference_order_cost"] = row
except Exception as e:  # (dev aid: the other rows are still worth printing)
    out["streaming_reference_order_cost"] = {"error": repr(e)}
print(json.dumps(out))
