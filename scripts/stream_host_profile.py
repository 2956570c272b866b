"""Host-side wall clock of every call of the streaming plan cycle (bench.py GpuStreamCycle.step), median over frames."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "G800S"
map_size, n_obs, _ = bench.WORKLOADS[wl]
box = bench.exploration_box(map_size)
frames = bench.streaming_frames(map_size, n_obs, 60, seed=42)
rng = np.random.default_rng(1042)
ctrl = bench.make_trajectories(rng, 64, 32, np.array(box[0]) + 0.5, np.array(box[1]) - 0.5)
cyc = bench.GpuStreamCycle(map_size, box, frames, ctrl, device=0)
for _ in range(8):
    cyc.step()
cyc.finish()
T = {}
def tick(name, t0):
    t1 = time.perf_counter()
    T.setdefault(name, []).append((t1 - t0) * 1e6)
    return t1
tot = []
for _ in range(40):
    img, pos, q = cyc.frames[cyc.k % len(cyc.frames)]
    cyc.k += 1
    m = cyc.map
    ts = t = time.perf_counter()
    fused = m.inputDepthImage(img, pos, q) > 0
    t = tick("inputDepthImage", t)
    cyc.ff.searchFrontiersBegin()
    t = tick("searchFrontiersBegin", t)
    if fused:
        lo, hi = m.getLocalBound()
        t = tick("getLocalBound", t)
        m.clearAndInflateLocalMap()
        t = tick("inflate", t)
        m.updateESDF3d()
        t = tick("updateESDF3d", t)
    cyc.dev_problem.eval()
    t = tick("bspline eval", t)
    cyc.n_clusters = cyc.ff.searchFrontiersEnd()
    t = tick("searchFrontiersEnd", t)
    cyc.ff.commit()
    t = tick("commit", t)
    tot.append((t - ts) * 1e6)
cyc.finish()
for k, v in T.items():
    print("%-22s median %7.1f us  (n=%d)" % (k, float(np.median(v)), len(v)))
print("%-22s median %7.1f us" % ("whole step", float(np.median(tot))))
