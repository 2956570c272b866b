"""Regenerates tests/golden/sdf_map_layout.txt: sizeof / offsetof lines of MapParam, MapData and SDFMap printed by
tests/golden/sdf_map_layout_probe.cpp compiled against the REFERENCE's plan_env/sdf_map.h (this container only:
needs /root/reference).  tests/test_abi_cpu.py::test_sdf_map_header_layout_is_the_references compares the facade's
header with this file and, where the reference checkout is present, the reference with it -- read-only."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_INC = "/root/reference/fuel_planner/plan_env/include"


def main():
    if not os.path.exists(os.path.join(REF_INC, "plan_env", "sdf_map.h")):
        sys.exit("no reference checkout at /root/reference")
    shims = [os.path.join(ROOT, "oracle", "ref_build", "shim_ros"), os.path.join(ROOT, "compat")]
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "probe_ref")
        cmd = ["g++", "-std=c++14", "-O0", os.path.join(ROOT, "tests", "golden", "sdf_map_layout_probe.cpp"), "-o", exe]
        for d in [REF_INC] + shims:
            cmd += ["-I", d]
        subprocess.check_call(cmd)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(ROOT, "tests", "golden", "sdf_map_layout.txt"), "w") as f:
        f.write(out if out.endswith("\n") else out + "\n")
    print("wrote tests/golden/sdf_map_layout.txt (%d lines)" % len(out.splitlines()))


if __name__ == "__main__":
    main()
