"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/fuelmi.h declares, the ctypes mirrors match the C struct layouts, and without a GPU the
product fails loudly (there is no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fuelmi.h")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import fuel_amd
    return fuel_amd


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fuelmi_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_all_exported(built):
    names = declared_functions()
    assert len(names) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH]).decode()
    exported = set(re.findall(r" T (fuelmi_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in include/fuelmi.h but not exported: %s" % missing


def test_python_binding_covers_header(built):
    from fuel_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared_functions()
    L = built.lib()
    assert L.fuelmi_version().startswith(b"fuelmi")


def test_ctypes_struct_layouts_match_c(built, tmp_path):
    from fuel_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fuelmi.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fuelmi_map_cfg), '
                    'sizeof(fuelmi_map_info), sizeof(fuelmi_frontier_cfg), sizeof(fuelmi_bspline_cfg), '
                    'sizeof(fuelmi_bspline_batch), offsetof(fuelmi_map_cfg, device), '
                    'offsetof(fuelmi_bspline_batch, view_idx));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.MapCfg), C.sizeof(_lib.MapInfo), C.sizeof(_lib.FrontierCfg), C.sizeof(_lib.BsplineCfg),
            C.sizeof(_lib.BsplineBatch), _lib.MapCfg.device.offset, _lib.BsplineBatch.view_idx.offset]
    assert got == want


def test_header_is_plain_c(tmp_path):
    prog = tmp_path / "c.c"
    prog.write_text('#include "fuelmi.h"\nint main(void){return FUELMI_OK;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(prog), "-o", str(tmp_path / "c.o")])


def test_no_gpu_means_loud_failure(built):
    L = built.lib()
    if L.fuelmi_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(built.FuelmiError) as e:
        built.SDFMap((4.0, 4.0, 2.0))
    assert "no CPU fallback" in str(e.value) or "-2" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """Nothing under fuel_amd/ or include/ may import, include or link the oracle."""
    bad = []
    for base in ("fuel_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|fuel_oracle|libfuel_oracle|oracle/", txt):
                        # comments that merely mention the oracle are fine in docs, not in code
                        for line in txt.splitlines():
                            s = line.strip()
                            if re.search(r"(from|import)\s+oracle|#include.*oracle|libfuel_oracle", s) and \
                                    not s.startswith(("#", "//", "*", '"""')):
                                bad.append((f, s))
    assert not bad, bad


def _layout_probe(tmp_path, header_dirs, name):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++14", "-w", "-o", exe, os.path.join(root, "tests", "golden", "sdf_map_layout_probe.cpp")]
    for d in header_dirs:
        cmd += ["-I", d]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    return subprocess.run([exe], capture_output=True, text=True, timeout=60, check=True).stdout.splitlines()


def test_sdf_map_header_layout_is_the_references(tmp_path):
    """SURVEY 8(b) "header-level ABI": MapParam, MapData and SDFMap's data members in the facade's plan_env/sdf_map.h
    have the reference's fields at the reference's offsets (plan_env/include/plan_env/sdf_map.h:74-125) -- the inline
    getters of that header are compiled into the callers.  One probe, compiled against each header with the same
    Eigen / ROS / PCL stand-ins; the reference's output is also kept as a golden file so that the comparison runs where
    the reference checkout is absent."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shims = [os.path.join(root, "oracle", "ref_build", "shim_ros"), os.path.join(root, "compat")]
    golden = os.path.join(root, "tests", "golden", "sdf_map_layout.txt")
    ref_inc = "/root/reference/fuel_planner/plan_env/include"
    assert os.path.exists(golden), "tests/golden/sdf_map_layout.txt is missing (regenerate: tests/golden/make_layout_golden.py)"
    if os.path.exists(os.path.join(ref_inc, "plan_env", "sdf_map.h")):
        # read-only: the reference's probe must EQUAL the committed golden; a drift of the reference checkout,
        # the probe or the shims fails here instead of silently rewriting a tracked file
        ref_live = _layout_probe(tmp_path, [ref_inc] + shims, "probe_ref")
        assert open(golden).read().splitlines() == ref_live, \
            "the reference header's layout differs from tests/golden/sdf_map_layout.txt (tests/golden/make_layout_golden.py rewrites it)"
    ref = open(golden).read().splitlines()
    ours = _layout_probe(tmp_path, [os.path.join(root, "fuel_amd", "facade"), os.path.join(root, "include")] + shims, "probe_ours")
    assert len(ref) == len(ours) and len(ref) > 50
    for a, b in zip(ref, ours):
        if a.startswith("SDFMap size"):
            # the drop-in appends exactly one pointer behind the reference's last member
            assert int(b.split()[-1]) == int(a.split()[-1]) + 8, (a, b)
        else:
            assert a == b, (a, b)


def test_hardware_queue_default_is_explicit():
    """include/fuelmi.h fuelmi_init / fuelmi_hw_queues / fuelmi_hw_queues_state: loading
    libfuelmi.so no longer touches the environment; fuelmi_init() (which fuel_amd.lib() calls right after the dlopen, before
    anything initialises HIP) puts GPU_MAX_HW_QUEUES=16 there unless the environment already decides, and says which of
    the two happened; the load-time constructor is opt-in (FUELMI_SET_HW_QUEUES)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    raw = ("import ctypes, os; L = ctypes.CDLL(os.path.join(%r, 'fuel_amd', 'libfuelmi.so')); "
           "print(os.environ.get('GPU_MAX_HW_QUEUES', 'unset'), L.fuelmi_hw_queues(), L.fuelmi_hw_queues_state())" % root)
    via = ("import os, fuel_amd; L = fuel_amd.lib(); "
           "print(os.environ.get('GPU_MAX_HW_QUEUES', 'unset'), L.fuelmi_hw_queues(), L.fuelmi_hw_queues_state())")

    def run(code, **extra):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FUELMI_SET_HW_QUEUES")}
        env.update(extra)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        return p.stdout.strip().splitlines()[-1].split()

    UNINIT, SET, ENV = 0, 1, 2
    # (os.environ is Python's snapshot from start-up: a setenv made by the C library does not show there -- the C side's
    # view is what fuelmi_hw_queues() returns)
    assert run(raw)[1:] == ["4", str(UNINIT)]            # a bare dlopen changes nothing
    assert run(raw, FUELMI_SET_HW_QUEUES="1")[1:] == ["16", str(SET)]   # opt-in constructor
    assert run(via)[1:] == ["16", str(SET)]              # the Python binding calls fuelmi_init() itself
    assert run(via, GPU_MAX_HW_QUEUES="8")[1:] == ["8", str(ENV)]       # the environment decides
