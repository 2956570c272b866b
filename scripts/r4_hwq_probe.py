"""does the streaming cycle slow down when other maps / finders (and their streams) are alive in the process?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, fuel_amd
map_size_s, n_obs_s, _ = bench.WORKLOADS["G800S"]
box_s = bench.exploration_box(map_size_s)
frames_s = bench.streaming_frames(map_size_s, n_obs_s, 124, seed=42)
ctrl_s = bench.make_trajectories(np.random.default_rng(1042), 64, 32, np.array(box_s[0]) + 0.5, np.array(box_s[1]) - 0.5)
def run(tag):
    cyc = bench.GpuStreamCycle(map_size_s, box_s, frames_s, ctrl_s, device=0, reference_order=0)
    cyc.run_native(20); cyc.finish()
    sec = cyc.run_native(100); cyc.finish()
    print(tag, "ms/frame %.4f" % (sec / 100 * 1e3)); sys.stdout.flush()
    cyc.close()
run("alone")
keep = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    m = fuel_amd.SDFMap((10.0, 10.0, 5.0)); f = fuel_amd.FrontierFinder(m, cluster_min=10)
    m.setUpdatedBox((-4, -4, 0), (4, 4, 2)); f.searchFrontiers(); f.reset(); m.setUpdatedBox((-4, -4, 0), (4, 4, 2)); f.searchFrontiers()
    keep.append((m, f))
    run("with %d idle maps+finders alive" % (k + 1))
