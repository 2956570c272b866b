"""dev aid: host-side time of every call of one plan cycle (perf_counter around the ctypes calls)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

map_size, box, occ, ctrl, n_known = bench.build_inputs("G400", seed=42)
cyc = bench.GpuCycle(map_size, box, occ, ctrl, device=0)
for _ in range(5):
    cyc.step()
cyc.finish()
m, ff = cyc.map, cyc.ff
calls = [("reset", ff.reset), ("setUpdatedBox", lambda: m.setUpdatedBox(box[0], box[1])),
         ("searchBegin", ff.searchFrontiersBegin), ("inflate", m.clearAndInflateLocalMap),
         ("esdf", m.updateESDF3d), ("bspline", cyc.dev_problem.eval), ("searchEnd", ff.searchFrontiersEnd)]
acc = {k: 0.0 for k, _ in calls}
N = 200
t_all = time.perf_counter()
for _ in range(N):
    for k, fn in calls:
        t0 = time.perf_counter()
        fn()
        acc[k] += time.perf_counter() - t0
t_all = time.perf_counter() - t_all
for k, _ in calls:
    print("%-14s %7.1f us" % (k, acc[k] / N * 1e6))
print("cycle %.1f us" % (t_all / N * 1e6))
print("clusters", [len(c) for c in ff.clusters(0)])
