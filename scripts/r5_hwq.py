"""round 5: the library against the hardware-queue default.  (a) streaming frame of map 0 alone and with 4 idle maps +
finders alive in the process; (b) ten optimiser threads, ten solves -- each at the environment the process was started
with (GPU_MAX_HW_QUEUES unset: the library's fuelmi_init() default applies; GPU_MAX_HW_QUEUES=4: the runtime's 4)."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, fuel_amd
import helpers
print("GPU_MAX_HW_QUEUES as seen by the runtime:", fuel_amd.lib().fuelmi_hw_queues()); sys.stdout.flush()
map_size_s, n_obs_s, _ = bench.WORKLOADS["G800S"]
box_s = bench.exploration_box(map_size_s)
frames_s = bench.streaming_frames(map_size_s, n_obs_s, 124, seed=42)
ctrl_s = bench.make_trajectories(np.random.default_rng(1042), 64, 32, np.array(box_s[0]) + 0.5, np.array(box_s[1]) - 0.5)
def run(tag):
    cyc = bench.GpuStreamCycle(map_size_s, box_s, frames_s, ctrl_s, device=0, reference_order=0)
    cyc.run_native(20); cyc.finish()
    best = 1e9
    for _ in range(3):
        sec = cyc.run_native(30); cyc.finish()
        best = min(best, sec / 30 * 1e3)
    print(tag, "ms/frame %.4f" % best); sys.stdout.flush()
    cyc.close()
    return best
solo = run("alone")
keep = []
for k in range(4):
    m = fuel_amd.SDFMap((10.0, 10.0, 5.0)); f = fuel_amd.FrontierFinder(m, cluster_min=10)
    m.setUpdatedBox((-4, -4, 0), (4, 4, 2)); f.searchFrontiers(); f.reset(); m.setUpdatedBox((-4, -4, 0), (4, 4, 2)); f.searchFrontiers()
    keep.append((m, f))
crowd = run("with 4 idle maps+finders alive")
print("ratio %.2f" % (crowd / solo))
# (b) ten optimiser threads on one map
om, _, _, box = helpers.explored_oracle_map((20.0, 20.0, 5.0), 60, 40)
gm = fuel_amd.SDFMap(tuple(om.cfg.map_size), box[0], box[1])
gm.uploadOccupancy(om.occ)
lo, hi = helpers.full_box(om.nvox)
gm.setLocalBound(lo, hi); gm.clearAndInflateLocalMap(); gm.updateESDF3d(); gm.synchronize()
rng = np.random.default_rng(3)
probs = []
for t in range(10):
    ctrl = helpers.make_trajectories(rng, 1, 24, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
    x, ptd, st, en = helpers.bspline_inputs(ctrl, 0.175, True)
    probs.append(fuel_amd.BsplineBatchProblem(x, 24, fuel_amd.NORMAL_PHASE | fuel_amd.MINTIME, ptd, st, en, 3, 3, 0.175))
opts = []
for t in range(10):
    o = fuel_amd.BsplineOptimizer(); o.setEnvironment(gm); opts.append(o)
for t in range(10):
    opts[t].optimize(probs[t])
t0 = time.perf_counter(); opts[0].optimize(probs[0]); one = time.perf_counter() - t0
def solve(t):
    opts[t].optimize(probs[t])
best = 1e9
for rep in range(5):
    th = [threading.Thread(target=solve, args=(t,)) for t in range(10)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    best = min(best, time.perf_counter() - t0)
print("one solve %.3f ms; ten threads, ten solves %.3f ms" % (one * 1e3, best * 1e3))
