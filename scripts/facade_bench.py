#!/usr/bin/env python3
"""Times the streaming plan cycle through the C++ facade (fuel_amd/facade/facade_bench: the reference's class
interfaces, host mirrors on / off) beside the same sequence at the C-ABI.  Frames: 640x480 depth frames of a
seeded synthetic world rendered to point clouds (skip 2), as MapROS hands them to SDFMap::inputPointCloud.
    python scripts/facade_bench.py [--map G800S|G400S] [--frames 30]  -> one JSON line
    python scripts/facade_bench.py --fullbox G400                      -> the headline cycle through the facade"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAPS = {"G800S": ((80.0, 80.0, 20.0), 600), "G400S": ((40.0, 40.0, 10.0), 150), "G100S": ((10.0, 10.0, 5.0), 12)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="G800S", choices=sorted(MAPS))
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--fullbox", default=None, choices=("G400", "G200", "G100"),
                    help="BASELINE's headline cycle (full-box inflate + ESDF + full search) through the facade classes")
    args = ap.parse_args()
    if args.fullbox:
        import bench
        map_size, box, occ, _, _ = bench.build_inputs(args.fullbox, seed=42, n_traj=1)
        with tempfile.TemporaryDirectory() as td:
            scen, occf = os.path.join(td, "scen.bin"), os.path.join(td, "occ.bin")
            with open(scen, "wb") as f:
                f.write(struct.pack("10d", *map_size, *box[0], *box[1], 100.0))
                f.write(struct.pack("i", 0))
            np.ascontiguousarray(occ, np.float64).tofile(occf)
            exe = os.path.join(ROOT, "fuel_amd", "facade", "facade_bench")
            out = subprocess.run([exe, scen, str(args.repeat), "fullbox", occf], capture_output=True, text=True, timeout=900)
            sys.stderr.write(out.stderr[-2000:])
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else '{"error": "no output"}')
        return
    from fuel_amd import synth
    map_size, n_obs = MAPS[args.map]
    w = synth.World.for_map_size(map_size)
    truth = w.world(42, n_obs)
    org = (-map_size[0] / 2.0, -map_size[1] / 2.0, -1.0)
    box = ((org[0] + 1.0, org[1] + 1.0, 0.0), (-org[0] - 1.0, -org[1] - 1.0, max(0.8 * map_size[2] - 1.0, 1.0)))
    frames = []
    n_try = 6 * args.frames
    for k in range(n_try):
        pose = w.camera(truth, 49, k, n_try, 0.7)
        pts = w.render(truth, pose, 640, 480, 2, 2)
        if len(pts) < 1000:
            continue
        frames.append((pts, pose[:3].copy()))
        if len(frames) == args.frames:
            break
    with tempfile.TemporaryDirectory() as td:
        scen = os.path.join(td, "scen.bin")
        with open(scen, "wb") as f:
            f.write(struct.pack("10d", *map_size, *box[0], *box[1], 100.0))
            f.write(struct.pack("i", len(frames)))
            for pts, cam in frames:
                f.write(struct.pack("i", len(pts)))
                f.write(struct.pack("3d", *cam))
                f.write(np.ascontiguousarray(pts, np.float32).tobytes())
        exe = os.path.join(ROOT, "fuel_amd", "facade", "facade_bench")
        out = subprocess.run([exe, scen, str(args.repeat)], capture_output=True, text=True, timeout=900)
        sys.stderr.write(out.stderr[-2000:])
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else '{"error": "no output"}')


if __name__ == "__main__":
    main()
